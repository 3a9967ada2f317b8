"""A small exact-match seeder that writes the seed format of row f3, for the plumbing run (BASELINE.json configs[0]).

In the reference the seeds come from the external `namfinder` binary (/root/reference/modules/seed_wrapper.py:73-101,
called at uLTRA:364) and north_star keeps it that way ("namfinder seeding stays"); the binary is not part of this
image.  This module is a stand-in so that the bundled SIRV reads can travel reads -> seeds -> NAM chaining -> MAM
stage here: maximal exact matches of at least `min_len` bases between a read (both strands) and the exon parts /
flanks (`ref_part_sequences`, uLTRA:339-344), written exactly as namfinder writes its hits.  It does not claim to equal
namfinder's strobemer NAMs; everything downstream only needs seeds in that format.  Off the hot path, plain Python.
"""
import gzip
from array import array

from .align import remove_read_polyA_ends

_COMP = str.maketrans("ACGTUNacgtun", "TGCAANtgcaan")


def reverse_complement(seq):
    return seq.translate(_COMP)[::-1]


def part_name(key):
    a = array("L")
    a.frombytes(key)
    return "{0}^{1}^{2}".format(a[0], a[1], a[2])


class MemIndex(object):
    def __init__(self, ref_part_sequences, min_len=17):
        self.k = int(min_len)
        self.names, self.seqs, self.table = [], [], {}
        for key, seq in ref_part_sequences.items():
            pid = len(self.names)
            self.names.append(part_name(key) if isinstance(key, (bytes, bytearray)) else str(key))
            seq = seq.upper()
            self.seqs.append(seq)
            for j in range(len(seq) - self.k + 1):
                self.table.setdefault(seq[j:j + self.k], []).append((pid, j))

    def mems(self, read):
        """[(ref name, ref pos, read pos, length)], positions 1-based, ordered by read position."""
        read = read.upper()
        k, out = self.k, []
        for i in range(len(read) - k + 1):
            for pid, j in self.table.get(read[i:i + k], ()):
                ref = self.seqs[pid]
                if i > 0 and j > 0 and read[i - 1] == ref[j - 1]:
                    continue                      # not left-maximal: reported from an earlier read position
                n = k
                while i + n < len(read) and j + n < len(ref) and read[i + n] == ref[j + n]:
                    n += 1
                out.append((self.names[pid], j + 1, i + 1, n))
        return out


def write_seed_file(path, reads, ref_part_sequences, min_len=17, reduce_read_polyA=20, compress=True):
    """reads: iterable of (acc, seq).  Seeds are computed on the read with its poly-A runs shortened to 5 bases, as the
    reference prepares reads for the seeder (uLTRA:354-359)."""
    index = MemIndex(ref_part_sequences, min_len)
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if compress else (lambda p: open(p, "wb"))
    n = 0
    with opener(path) as fh:
        for acc, seq in reads:
            trimmed, _ = remove_read_polyA_ends(seq, None, reduce_read_polyA, 5)
            for header, s in ((acc, trimmed), (acc + " Reverse", reverse_complement(trimmed))):
                fh.write(">{0}\n".format(header).encode())
                for name, rpos, qpos, ln in index.mems(s):
                    fh.write("{0}\t{1}\t{2}\t{3}\n".format(name, rpos, qpos, ln).encode())
                    n += 1
    return n
