"""Synthetic coordinate-sorted BAM record streams (generator G1) through the C ABI.

`host_stream` runs the generator on the host; `Context.synth_device` runs the very same functions
(biod_amd/csrc/synth_record.h) in a gfx950 kernel straight into HBM, byte-identical by construction."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

# BASELINE.json configs (reads, profile, read length, contigs, contig length, seed)
CONFIGS = {
    "cfg1": dict(n_reads=10_000, n_ref=2, ref_len=50_000, read_len=100, profile=1, seed=1),
    "cfg2": dict(n_reads=10_000_000, n_ref=8, ref_len=6_250_000, read_len=150, profile=0, seed=2),
    "cfg3": dict(n_reads=100_000_000, n_ref=8, ref_len=62_500_000, read_len=150, profile=1, seed=3),
    "cfg5": dict(n_reads=200_000, n_ref=1, ref_len=200_000_000, read_len=(10_000, 50_000), profile=2, seed=5),
}


def make_cfg(n_reads, n_ref, ref_len, read_len, profile, seed, first_ref=0) -> L.SynthCfg:
    lo, hi = (read_len, read_len) if isinstance(read_len, int) else read_len
    return L.SynthCfg(seed=seed, n_reads=n_reads, n_ref=n_ref, ref_len=ref_len, len_min=lo, len_max=hi, profile=profile, first_ref=first_ref)


def scaled(name: str, n_reads: int, first_ref: int = 0) -> L.SynthCfg:
    """A configuration with the same coverage depth and read model as `name`, at `n_reads` reads."""
    c = dict(CONFIGS[name])
    f = n_reads / c["n_reads"]
    c["ref_len"] = max(int(c["ref_len"] * f), 200_000 if c["profile"] == 2 else 20_000)
    c["n_reads"] = n_reads
    return make_cfg(first_ref=first_ref, **c)


def _host_chunk(cfg: L.SynthCfg, first: int, count: int) -> np.ndarray:
    n = C.c_uint64()
    L.check(L.lib().bgp_synth_host(C.byref(cfg), first, count, None, 0, C.byref(n)))
    buf = np.zeros(n.value, np.uint8)
    L.check(L.lib().bgp_synth_host(C.byref(cfg), first, count, buf.ctypes.data, n.value, C.byref(n)))
    return buf


def host_stream_np(cfg: L.SynthCfg, first: int = 0, count: int | None = None) -> np.ndarray:
    """Records [first, first+count) generated on the host (threads over index ranges: records are
    pure functions of their index, and ctypes releases the GIL)."""
    count = cfg.n_reads - first if count is None else count
    if count < 50_000:
        return _host_chunk(cfg, first, count)
    import os
    from concurrent.futures import ThreadPoolExecutor
    nt = min(os.cpu_count() or 1, 32)
    step = (count + nt * 4 - 1) // (nt * 4)
    parts = [(first + o, min(step, count - o)) for o in range(0, count, step)]
    with ThreadPoolExecutor(nt) as ex:
        bufs = list(ex.map(lambda p: _host_chunk(cfg, p[0], p[1]), parts))
    return np.concatenate(bufs)


def host_stream(cfg: L.SynthCfg, first: int = 0, count: int | None = None) -> bytes:
    return host_stream_np(cfg, first, count).tobytes()


def reads_per_contig(cfg: L.SynthCfg) -> int:
    """synth_record.h: reads_per_contig -- record i lies on local contig i // this."""
    return (int(cfg.n_reads) + int(cfg.n_ref) - 1) // int(cfg.n_ref)


def max_span(cfg: L.SynthCfg) -> int:
    """synth_record.h: max_span -- no read of the configuration covers more reference positions than this."""
    if cfg.profile == 0:
        return int(cfg.len_max)
    if cfg.profile == 1:
        return int(cfg.len_max) + 5000 + 32
    return int(cfg.len_max) + int(cfg.len_max) // 2 + 64
