"""BASELINE.json configs[3]: the whole-genome workload -- the 25 hg38 primary contigs at 30x (config-2
read model), a FIXED input dealt to the ranks contig by contig (strong scaling).

What is sharded is the reference's serial contig loop (src/api/coverage.rs:229-234): contigs are
independent, so every rank generates, admits, uploads and keeps resident only the contigs the
longest-processing-time-first deal gives it (coverage.lpt_assignment on the contig lengths, the same
deal on every rank), runs them on its own GPU with no data-path collective, and the per-contig summary
records are exchanged with ONE all_gather taken straight from the engines' resident cl_contig_summary
records (cl_device_summary, include/callable_loci.h) -- RCCL over xGMI with the nccl backend.

Used by bench.py (--gpus N > 1, or --workload wgs) and by the tests (scaled-down genomes).
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import synth
from .callable_loci import CallableOptions, Engine
from .coverage import ContigOutcome, device_summary_tensor, engine_process_contig_runs, lpt_assignment

SUMMARY_WORDS = 14          # u64 words of cl_contig_summary


def genome(scale: float = 1.0, min_len: int = 1000):
    """[(tid, name, length)] of the hg38 primary contigs, lengths scaled (tests use 1/16 and less)."""
    return [(tid, name, max(min_len, int(L * scale)) if scale != 1.0 else L)
            for tid, (name, L) in enumerate(synth.HG38_PRIMARY)]


def deal(contigs, world: int) -> List[int]:
    """rank of every contig: LPT on the lengths (aligned bases are proportional to them at fixed depth)."""
    return lpt_assignment([L for _, _, L in contigs], world)


def make_contig(tid: int, L: int, depth: float):
    """The contig's synthetic records and reference bases (seeded by config 4 and the tid)."""
    seed = synth.seed_for(4, tid)
    return synth.short_read_contig(L, depth, seed), synth.make_reference(L, seed)


@dataclass
class ResidentContig:
    tid: int
    name: str
    length: int
    engine: Engine
    outcome: ContigOutcome              # first pass through the module API (admission, runs, summary)
    n_reads: int
    aligned_bases: int
    dev_summary: object = None          # torch view of the engine's resident cl_contig_summary
    first_pass_s: float = 0.0
    gen_s: float = 0.0


@dataclass
class ResidentShard:
    """This rank's share of the genome, every contig resident in HBM on its own engine context.  The contexts
    enqueue in turn on four streams -- the caller's and three of the shard's own: contigs are independent, so the short
    tail of one contig's pass (two small kernels, the drain of its pileup kernel) runs beside the next contig's pileup
    kernel; `step()` makes the caller's stream wait for the others, so whatever follows on it (the gather of the
    summaries, a synchronize) sees every contig done."""
    rank: int
    world: int
    contigs: list                       # the whole genome [(tid, name, L)]
    rank_of: List[int]
    mine: List[ResidentContig] = field(default_factory=list)
    side_stream: object = None          # torch.cuda.Stream: the engines of every second contig run on it
    side_streams: list = field(default_factory=list)

    @property
    def bases(self) -> int:
        return sum(c.length for c in self.mine)

    def step(self):
        """One pass of the device path over every resident contig of this rank (asynchronous)."""
        for c in self.mine:
            c.engine.contig_run()
        if self.side_stream is not None:
            import torch
            for st in self.side_streams:
                torch.cuda.current_stream().wait_event(st.record_event())

    def sync(self):
        for c in self.mine:
            c.engine.sync()

    def summary_rows(self, device):
        """[per_rank, 14] int64 on `device`: the resident summary records of this rank's contigs in deal order,
        copied device to device (rows beyond the rank's count are -1)."""
        import torch
        per_rank = max(1, max(self.rank_of.count(r) for r in range(self.world)))
        on_dev = str(device).startswith("cuda")
        rows = torch.full((per_rank, SUMMARY_WORDS), -1, dtype=torch.int64, device=device if on_dev else "cpu")
        if self.mine:
            src = torch.stack([c.dev_summary for c in self.mine])       # HBM -> HBM, on the current stream
            rows[:len(self.mine)] = src if on_dev else src.cpu()
        return rows

    def gather_device(self, device, group=None):
        """The one exchange of the path: all_gather of the summary rows (asynchronous on the current stream with
        the nccl backend).  Returns the per-rank row tensors."""
        import torch
        import torch.distributed as dist
        rows = self.summary_rows(device)
        if self.world > 1:
            out = [torch.empty_like(rows) for _ in range(self.world)]
            dist.all_gather(out, rows, group=group)
            return out
        return [rows]

    def parse(self, gathered) -> Dict[int, List[int]]:
        """{tid: the 14 words of its cl_contig_summary} from the gathered rows (same deal on every rank)."""
        table = {}
        for r, g in enumerate(gathered):
            tids = [t for (t, _, _), rr in zip(self.contigs, self.rank_of) if rr == r]
            for t, row in zip(tids, g.cpu().tolist()):
                table[t] = row
        return table

    def gather_summaries(self, device, group=None) -> Dict[int, List[int]]:
        return self.parse(self.gather_device(device, group))

    def close(self):
        for c in self.mine:
            c.engine.close()
        self.mine = []


def build_shard(rank: int, world: int, device_id: int, options: CallableOptions, depth: float = 30.0,
                scale: float = 1.0, stream: int = 0, gen_threads: int = 4, log=None,
                keep_records: Optional[dict] = None, two_streams: bool = True) -> ResidentShard:
    """Generate this rank's contigs (`gen_threads` at a time on host threads: numpy releases the GIL), push each
    through the module API once (admission + H2D + kernels + D2H of the runs) and keep it resident.
    `keep_records`: a dict that receives {tid: (records, ref)} (tests compare against the oracle)."""
    contigs = genome(scale)
    rank_of = deal(contigs, world)
    shard = ResidentShard(rank, world, contigs, rank_of)
    mine = [(t, nm, L) for (t, nm, L), r in zip(contigs, rank_of) if r == rank]
    streams = [stream]
    n_side = int(os.environ.get("DUT_WGS_SIDE_STREAMS", "3"))               # measured: 1 stream 5.40 ms per step, 2: 5.10, 4: 4.97; 0: the caller's only
    if two_streams and n_side > 0 and len(mine) > 1:
        import torch
        shard.side_streams = [torch.cuda.Stream(device=device_id) for _ in range(n_side)]
        shard.side_stream = shard.side_streams[0]
        streams += [st.cuda_stream for st in shard.side_streams]

    def gen(item):
        t0 = time.perf_counter()
        rec, ref = make_contig(item[0], item[2], depth)
        return rec, ref, time.perf_counter() - t0

    # Batches of `gen_threads` contigs: generated side by side, then -- with no generator running, so that the first
    # passes are timed on a quiet host like a real run's -- pushed through the module API one after the other.
    B = max(1, gen_threads)
    with ThreadPoolExecutor(B) as pool:
        for b0 in range(0, len(mine), B):
            batch = mine[b0:b0 + B]
            made = [f.result() for f in [pool.submit(gen, item) for item in batch]]
            for i, (tid, name, L) in enumerate(batch):
                (rec, ref, gen_s), made[i] = made[i], None           # the records are freed contig by contig
                eng = Engine(options, device_id, streams[len(shard.mine) % len(streams)])
                t0 = time.perf_counter()
                out = engine_process_contig_runs(eng, options, tid, name, L, rec, ref)
                first = time.perf_counter() - t0
                rc = ResidentContig(tid, name, L, eng, out, rec.n, int(rec.qual.shape[0]),
                                    dev_summary=device_summary_tensor(eng), first_pass_s=first, gen_s=gen_s)
                shard.mine.append(rc)
                if keep_records is not None:
                    keep_records[tid] = (rec, ref)
                if log:
                    log(f"[wgs r{rank}] {name}: {L} bp, {rec.n} reads, generated in {gen_s:.1f}s, first pass {first:.2f}s, "
                        f"{out.intervals.shape[0]} runs")
                del rec, ref
            del made
    return shard
