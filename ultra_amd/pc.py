"""Producer / consumer front of the aligner with the reference's interface (/root/reference/modules/pc.py:27-121):

    main(reads, seeds, outfile, args) -> tot_counts[8]        (called at uLTRA:382)

The reference starts one file_IO producer and args.nr_cores align_single consumers, each consumer a CPU worker with
its own copy of the index.  Here a consumer drives a GPU (ultra_amd.align.align_single), so the fan-out is one
consumer per GPU of the node - args.gpus, else every visible gfx950 device - each bound to its device through
LOCAL_RANK.  Reads are independent: no exchange between consumers.  The queue protocol is the
reference's: 1000-read batches [batch_id, [[acc, seq, qual, hits, hits_rc], ...]], a None sentinel at the end,
lists of SAM rows coming back, one 8-entry classification list per consumer.
"""
import multiprocessing as mp
import os
from queue import Empty

from . import seed_wrapper

READS_PER_QUEUE_BATCH = 1000     # pc.py:40


def readfq(fp):
    """FASTA / FASTQ records as (name, (seq, qual or None)); what help_functions.readfq yields (the reference
    vendors lh3's readfq; this is an independent reader with the same results on well-formed files)."""
    name, seq, qual, in_qual = None, [], [], False
    for line in fp:
        line = line.rstrip("\n")
        if in_qual:                                  # quality lines may start with '@' or '>': count, do not parse
            qual.append(line)
            if sum(len(q) for q in qual) >= sum(len(s) for s in seq):
                yield name, ("".join(seq), "".join(qual))
                name, seq, qual, in_qual = None, [], [], False
        elif line[:1] in (">", "@"):
            if name is not None:
                yield name, ("".join(seq), None)
            fields = line[1:].split()
            name, seq = (fields[0] if fields else ""), []
        elif line[:1] == "+" and name is not None:
            in_qual = True
        elif name is not None:
            seq.append(line)
    if name is not None:
        yield name, ("".join(seq), None)


def _drain(outfile, output_sam_buffer):
    n = 0
    while True:
        try:
            rows = output_sam_buffer.get(False)
        except Empty:
            return n
        outfile.write("".join(rows))
        n += len(rows)


def file_IO(input_queue, reads, seeds, output_sam_buffer, outfile_name):
    """Producer (pc.py:27-62): zips the reads with their seed records into 1000-read batches."""
    written = 0
    with open(outfile_name, "a") as outfile:
        batch, batch_id, n_reads = [], 1, 0
        with open(reads, "r") as fh:
            for (acc, (seq, qual)), (r_acc, hits, r_acc_rev, hits_rc) in zip(readfq(fh), seed_wrapper.read_seeds(seeds)):
                if acc != r_acc:
                    raise ValueError("read %r and seed record %r are out of step" % (acc, r_acc))
                batch.append([acc, seq, qual, hits, hits_rc])
                n_reads += 1
                if n_reads % READS_PER_QUEUE_BATCH == 0:
                    input_queue.put([batch_id, batch])
                    batch, batch_id = [], batch_id + 1
        input_queue.put((batch_id, batch))
        input_queue.put(None)
        written += _drain(outfile, output_sam_buffer)
    print('file_IO: Reading records done. Tot read:', n_reads)
    return written


def _consumer(process_id, device, input_queue, output_sam_buffer, counts_queue, args):
    os.environ["LOCAL_RANK"] = str(device)      # before anything touches HIP in this process
    from . import hwq
    # a spawned child: HIP is not up yet.  The benched queue count, or this process's share of what one GPU schedules at once
    hwq.export_hw_queues(processes_per_gpu=max(1, int(getattr(args, "consumers_per_gpu", 1) or 1)))
    from . import align
    align.align_single(process_id, input_queue, output_sam_buffer, counts_queue, args)


def main(reads, seeds, outfile, args):
    from . import lib as _lib
    n_gpus = int(getattr(args, "gpus", 0) or 0)
    if n_gpus <= 0:
        n_gpus = _lib.load().ultra_device_count()
    if n_gpus <= 0:
        raise RuntimeError("ultra_amd.pc.main: no gfx950 device visible (there is no CPU fallback)")
    ctx = mp.get_context("spawn")               # a forked child must not inherit an initialised HIP runtime
    manager = ctx.Manager()
    input_queue = manager.Queue(200)
    output_sam_buffer = manager.Queue()
    counts_queue = manager.Queue()
    producer = ctx.Process(target=file_IO, args=(input_queue, reads, seeds, output_sam_buffer, outfile))
    producer.start()
    # consumers per GPU (default 1): the stages behind the MAM stage are the reference's own Python, and a second
    # process on the same card runs its host part while the first one's batch is on the device
    per_gpu = max(1, int(getattr(args, "consumers_per_gpu", 1) or 1))
    workers = [ctx.Process(target=_consumer, args=(i, i // per_gpu, input_queue, output_sam_buffer, counts_queue, args))
               for i in range(n_gpus * per_gpu)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    # Exit codes first: with every consumer dead the producer blocks forever on the 200-slot input queue, and joining it
    # would hang the run instead of reporting the failure.  One dead consumer is an error too (its batches are lost).
    failed = [w.exitcode for w in workers if w.exitcode != 0]
    if failed:
        producer.terminate()
        producer.join()
        manager.shutdown()
        raise RuntimeError("ultra_amd.pc.main: a GPU worker ended with exit code %r" % failed[0])
    producer.join()
    if producer.exitcode != 0:
        manager.shutdown()
        raise RuntimeError("ultra_amd.pc.main: the file_IO producer ended with exit code %r" % producer.exitcode)
    with open(outfile, "a") as fh:
        _drain(fh, output_sam_buffer)
    tot_counts = [0, 0, 0, 0, 0, 0, 0, 0]
    while True:
        try:
            res = counts_queue.get(False)
        except Empty:
            break
        for i in range(len(res)):
            tot_counts[i] += res[i]
    return tot_counts
