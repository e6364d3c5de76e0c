#!/usr/bin/env python3
"""bench.py -- BAM-record decode -> CIGAR walk -> pileup on MI355X, BASELINE.json's metric.

One "step" = one full pass of the hot path over one synthetic coordinate-sorted BAM record stream that
is already resident in HBM: record framing (K0), fixed fields + end_position (K1), SEQ/QUAL/CIGAR
unpack (K2), reference-run keys (K3) and the per-position pileup (K4).  Nothing is cached between
steps (bgp_reset drops every result).  At N GPUs every rank processes its own shard of contigs
(weak scaling, no data-path collective) and the whole-BAM depth histogram is summed over RCCL.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def pmc_traffic(kernel, config, n_reads):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected separately, KB units, FETCH doubled for gfx950 as MI355X_MICROARCH.md prescribes).  Counters
    cannot be read from inside this process, so the number is only reported when the workload is the one the
    profile was taken on (cfg3, 100 M reads); otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_cfg3_100M.csv")
    if config != "cfg3" or n_reads != 100_000_000 or not os.path.exists(path):
        return None
    total = None
    for line in open(path):
        f = line.strip().split(",")
        if f and (f[0] == kernel or f[0].startswith(kernel + "<")):      # template instantiations of one timed region add up
            total = (total or 0) + int(f[4])
    return total


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"], help="BASELINE.json config (cfg3 = the 100M-read mixed-CIGAR BAM the metric is quoted on)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the config's own size, reduced only if HBM is short)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads in the CPU-baseline sample (0 = auto, about 10-30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true", help="split ONE --reads-sized BAM by contig over the ranks (strong scaling) instead of one full shard per rank")
    return ap.parse_args()


def usable_cores():
    """Host cores this process can really run on: the scheduler's affinity mask, cut by the cgroup CPU quota
    (the GPU boxes show 256 logical CPUs under a 16-CPU cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg_name, n_sample):
    """The oracle (a C restatement of BioD's algorithm -- BioD itself is D and cannot be built here)
    timed on this box's host cores on a bounded sample of the same workload."""
    import oracle_lib as O
    from biod_amd.synth import host_stream_np, scaled
    cores = usable_cores()
    if n_sample <= 0:
        probe = host_stream_np(scaled(cfg_name, 20_000 if cfg_name != "cfg5" else 60))
        n_probe = 20_000 if cfg_name != "cfg5" else 60
        t0 = time.perf_counter(); O.pileup_bench(probe, 1); dt = max(time.perf_counter() - t0, 1e-4)
        n_sample = int(min(max(20.0 / (dt / n_probe), n_probe), 3_000_000))      # ~20 CPU-seconds of pileup work
        if cfg_name != "cfg5":
            # every thread needs work: pileupChunks cuts 16 384 000-byte chunks (~56k reads); give each core two
            n_sample = int(min(max(n_sample, cores * 2 * 56_000), 32_000_000))
    stream = host_stream_np(scaled(cfg_name, n_sample))
    t0 = time.perf_counter(); rs = O.pileup_bench(stream[: len(stream)], 1) if n_sample <= 400_000 else None; ts = time.perf_counter() - t0
    t0 = time.perf_counter(); rp = O.pileup_bench(stream, cores); tp = time.perf_counter() - t0
    assert rp.status in (0, -9), rp.status
    out = {"value": round(n_sample / tp / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
           "columns_per_s": round(rp.n_columns / tp, 1),
           "sample": f"{n_sample} reads of {cfg_name}'s read model at its coverage depth ({len(stream) / 1e6:.0f} MB inflated), "
                     f"pileupChunks-style decomposition over {cores} threads (= usable cores: {os.cpu_count()} logical CPUs, affinity and cgroup quota applied), "
                     f"consumer touches coverage/n_starting_here/every base+quality"}
    if rs is not None:
        out["serial_value"] = round(n_sample / ts / 1e6, 4)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from biod_amd import _lib as L
    from biod_amd.pileup import Context
    from biod_amd.synth import CONFIGS, scaled

    if torch.cuda.device_count() and local >= torch.cuda.device_count():
        local %= torch.cuda.device_count()          # (only for dry runs of the N>1 path on a box with fewer GPUs)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        backend = os.environ.get("BGP_BENCH_BACKEND", "nccl")      # nccl = RCCL over xGMI; gloo only to dry-run the control flow
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n_reads = a.reads or CONFIGS[a.config]["n_reads"]
    free, _total = torch.cuda.mem_get_info(dev)
    need_per_read = 1000 if a.config != "cfg5" else 600_000       # stream + tables + arenas + columns, generous
    while n_reads * need_per_read > free * 0.9 and n_reads > 1000:
        n_reads //= 2
    from biod_amd.shard import strong_cfg, weak_cfg
    if a.strong:        # BASELINE.json config 4 literally: the same BAM split by contig over the ranks
        cfg = strong_cfg(a.config, n_reads, rank, world)
        n_reads = int(cfg.n_reads)
    else:               # default: every rank owns a full-size shard of its own contigs
        cfg = weak_cfg(a.config, n_reads, rank)

    ctx = Context(local)
    t0 = time.perf_counter()
    stream_bytes = ctx.synth_device(cfg)                           # synthetic stream generated straight into HBM
    ctx.flush()
    r = ctx.decode_device(0)
    sptr, slen = r.stream, r.stream_len
    gen_s = time.perf_counter() - t0
    hist = torch.zeros(L.BGP_DEPTH_BINS, dtype=torch.int64, device=dev)
    hip = C.CDLL("libamdhip64.so")

    def step():
        ctx.reset()
        ctx.attach_device_stream(sptr, slen)
        ctx.flush()                                                # K0 framing + K1 record table
        ctx.decode_device(L.BGP_FETCH_ALL)                         # K2 seq/qual/cigar arenas
        info = ctx.pileup_run()                                    # K3 + K4 dense columns
        if world > 1:                                              # the path's only exchange: whole-BAM depth histogram over RCCL
            hip.hipMemcpy(C.c_void_p(hist.data_ptr()), C.c_void_p(info.device_depth_hist), C.c_size_t(L.BGP_DEPTH_BINS * 8), 3)
            dist.all_reduce(hist)
        return info

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        info = step()
    ctx.timing_enable(True); ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        info = step()
    sync()
    dt = time.perf_counter() - t0
    timers = ctx.timing_get()
    ctx.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([info.n_columns, info.sum_coverage], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        job_columns, job_cov = int(tot[0].item()), int(tot[1].item())
        assert int(hist.sum().item()) > 0
    else:
        job_columns, job_cov = info.n_columns, info.sum_coverage

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        job_reads = n_reads * world      # (strong: ranks hold equal contig shares of the one BAM)
        # roofline of the dominant kernel (largest share of kernel time), algorithmic bytes per launch / mean launch time
        kt = {k: v for k, v in timers.items() if v["launches"]}
        dom = max(kt, key=lambda k: kt[k]["total_ms"])
        per_kernel = {k: {"ms_per_launch": round(v["total_ms"] / v["launches"], 4), "launches_per_step": v["launches"] // a.steps,
                          "algorithmic_GBps": round(v["algorithmic_bytes"] / v["launches"] / (v["total_ms"] / v["launches"] * 1e-3) / 1e9, 1) if v["total_ms"] > 0 else None}
                      for k, v in kt.items()}
        d = kt[dom]
        achieved = d["algorithmic_bytes"] / d["launches"] / (d["total_ms"] / d["launches"] * 1e-3) / 1e9
        dec_ms = sum(kt[k]["total_ms"] for k in kt if k.startswith(("k0", "k1", "k2"))) / a.steps
        out = {
            "metric": "Mreads/s decoded + piled up (BAM record decode -> CIGAR walk -> per-base pileup, input resident in HBM)",
            "value": round(job_reads / dt * a.steps / 1e6, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
            "config": {"workload": f"{a.config}: {n_reads} reads/GPU synthetic coordinate-sorted BAM record stream "
                                   f"({'mixed M/I/D/S/N CIGAR' if cfg.profile == 1 else 'M-only CIGAR' if cfg.profile == 0 else 'long indel-heavy reads'}, "
                                   f"l_seq {cfg.len_min}-{cfg.len_max}, {cfg.n_ref} contigs x {cfg.ref_len} bp per GPU)",
                       "reads_per_gpu": n_reads, "inflated_bytes_per_gpu": stream_bytes, "parallelism": f"contig-sharded x{world}",
                       "step": "K0 framing + K1 record table + K2 seq/qual/cigar unpack + K3 run keys + K4 pileup columns"},
            "pileup_columns_per_s": round(job_columns / dt * a.steps, 1),
            "columns_per_step": job_columns, "sum_coverage_per_step": job_cov,
            "decode_Mreads_per_s_kernel_only": round(n_reads / (dec_ms * 1e-3) / 1e6, 2) if dec_ms else None,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": d["algorithmic_bytes"] // d["launches"],
                         "traffic": pmc_traffic(dom, a.config, n_reads)},
            "kernels": per_kernel,
            "setup_s": round(gen_s, 2),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config, a.cpu_sample_reads)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
