"""Seed input contract of the hot path: the text format `namfinder` writes and
/root/reference/modules/seed_wrapper.py:139-179 (`read_seeds`) streams to the aligner.  The seeder itself stays
what it is (north_star: "namfinder seeding stays"); this module reads its output with the reference's
semantics and can write the same format, which is how tests and the bench feed synthetic seeds through the
real entry point.

    >read_acc                                   forward record
    <ref_id> <ref_pos> <read_pos> <len>         one hit per line, whitespace separated, positions 1-based,
    ...                                         ref_id = "chrid^start^stop" of the exon part / flank
    >read_acc Reverse                           record of the reverse-complemented read (header contains "Reverse")
    ...
"""
import gzip


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def _blocks(fh):
    """(header or None, hit lines) per header-delimited block of a seed file; None = lines before any header."""
    header, lines = None, []
    for raw in fh:
        line = raw.decode('utf-8')
        if line[0] == '>':
            yield header, lines
            header, lines = line[1:].strip(), []
        else:
            lines.append(line)
    yield header, lines


def read_seeds(seeds):
    """Yields (acc, hits, acc_rev, hits_rc) per read, hits = the raw hit lines: the stream the reference's
    read_seeds produces (seed_wrapper.py:139-179).  A read is handed over at the first header that follows both its
    forward and its "Reverse" block (or at the end of the file); until then blocks accumulate, so a repeated forward
    header only renames the pending read, and hit lines before the first header belong to the first read's forward
    strand.  Plain text is accepted besides gzip."""
    pending = {"fw": '', "rc": ''}
    hits = {"fw": [], "rc": []}
    with _open(seeds) as fh:
        for header, lines in _blocks(fh):
            if header is None:
                hits["fw"] += lines
                continue
            if pending["fw"] and pending["rc"]:
                yield pending["fw"], hits["fw"], pending["rc"], hits["rc"]
                pending, hits = {"fw": '', "rc": ''}, {"fw": [], "rc": []}
            strand = "rc" if 'Reverse' in header else "fw"
            pending[strand] = header
            hits[strand] += lines
    if pending["fw"] and pending["rc"]:
        yield pending["fw"], hits["fw"], pending["rc"], hits["rc"]


def format_hit(mem):
    """Inverse of align.get_mems_from_input for one seed: Mem -> hit line (1-based part-local coordinates)."""
    part_start = int(mem.exon_part_id.split('^')[1])
    return "{0}\t{1}\t{2}\t{3}\n".format(mem.exon_part_id, mem.x - part_start + 1, mem.c + 1, mem.val)


def write_seeds(path, records, compress=True):
    """records: iterable of (acc, mems_fw, mems_rc), mems = Mem tuples (any order).  Writes the format above."""
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if compress else (lambda p: open(p, "wb"))
    with opener(path) as fh:
        for acc, fw, rc in records:
            fh.write(">{0}\n".format(acc).encode())
            for m in fw:
                fh.write(format_hit(m).encode())
            fh.write(">{0} Reverse\n".format(acc).encode())
            for m in rc:
                fh.write(format_hit(m).encode())
