"""`coverage` on files over several GPUs of one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m decodingustools_amd.coverage_mgpu sample.bam -r ref.fa -o callable_regions.bed [-L chr1 ...]

The flags are the reference CLI's (src/cli.rs:14-61).  Contigs are dealt to the ranks by
longest-processing-time-first on the index's mapped-read counts; every rank decodes and processes only its
own contigs; one all_gather of per-contig summary rows (RCCL over xGMI with --backend nccl) and a gather of
the run lists to rank 0, which writes the BED in header order and ./summary.json."""
import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="coverage_mgpu")
    ap.add_argument("bam_file")
    ap.add_argument("-r", "--reference", required=True)
    ap.add_argument("-o", "--output", default="callable_regions.bed")
    ap.add_argument("-s", "--summary", default="summary.html")
    ap.add_argument("-L", "--contig", action="append")
    ap.add_argument("--min-depth", type=int, default=4)
    ap.add_argument("--max-depth", type=int, default=500)
    ap.add_argument("--min-mapping-quality", type=int, default=10)
    ap.add_argument("--min-base-quality", type=int, default=20)
    ap.add_argument("--min-depth-for-low-mapq", type=int, default=10)
    ap.add_argument("--max-low-mapq", type=int, default=1)
    ap.add_argument("--max-low-mapq-fraction", type=float, default=0.1)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--summary-json", default="summary.json")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from .callable_loci import CallableOptions
    from .coverage import ApiError, coverage_files_sharded
    n_dev = torch.cuda.device_count()
    dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = CallableOptions(a.min_depth, a.max_depth, a.min_mapping_quality, a.min_base_quality, a.min_depth_for_low_mapq,
                          a.max_low_mapq, a.max_low_mapq_fraction)
    rc = 0
    try:
        coverage_files_sharded(a.bam_file, a.reference, a.output, a.summary_json, opt, a.contig, rank, world, dev,
                               coll_device="cuda" if (world > 1 and a.backend == "nccl") else "cpu", output_summary=a.summary)
    except (ApiError, OSError, RuntimeError) as e:
        print(f"Error: Analysis error: {e}", file=sys.stderr)
        rc = 1
    finally:
        if world > 1:
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
