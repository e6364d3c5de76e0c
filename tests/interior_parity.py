"""Interior-shard parity at full size (test infrastructure: used by tests/ and by bench.py AFTER its timed region).

With a whole configuration resident and piled up on the device (cfg3: 100 M reads, 29 GB of records), pick record
ranges [first, first + count) anywhere in the stream -- past 4 / 8 / 16 GiB of stream, arena and column offsets, across
contig boundaries, at the deepest column, at the very end --, generate exactly those records on the host (records are
pure functions of their index: biod_amd/csrc/synth_record.h), run the CPU oracle on them, and compare bit for bit

  (a) the device's bgp_record rows, the records' own stream bytes and their SEQ / QUAL / CIGAR arena slices,
  (b) every field of every column the range determines completely: the host stream carries a left halo of earlier
      reads wide enough that no read in front of it can reach the range's first position (the generator bounds a read's
      span), and one extra record on the right bounds the positions later reads cannot start before,
  (c) on request the per-read entries (K5) of a window inside the range.

Semantics compared: readrange.d:118-173, read.d:907-1003, pileup.d:115-134,175-222,345-397.
"""
from __future__ import annotations

import numpy as np

import oracle_lib as O
from biod_amd import _lib as L
from biod_amd.synth import host_stream, max_span, reads_per_contig

GIB = 1 << 30
FIELDS = ("ref_id", "pos", "end_pos", "bin_mq_nl", "flag_nc", "l_seq")


def _lower_bound(n, key_of, target):
    """Smallest record index i in [0, n] with key_of(i) >= target (keys are non-decreasing)."""
    lo, hi = 0, n
    while lo < hi:
        m = (lo + hi) // 2
        if key_of(m) >= target:
            hi = m
        else:
            lo = m + 1
    return lo


def plan_ranges(ctx, cfg, info, count, torch=None, dev=None, want=16):
    """[(label, first)] of the record ranges to check; every range has `count` records."""
    n = ctx.num_records()
    per = reads_per_contig(cfg)
    row = lambda i: ctx.decode_fetch_range(i, 1, 0)
    rec_off = lambda i: int(row(i).rec["rec_off"][0])
    seq_off = lambda i: int(row(i).seq_off[0])
    poskey = lambda i: (int(row(i).rec["ref_id"][0]) << 32) | int(row(i).rec["pos"][0])
    clamp = lambda f: max(0, min(n - count, f))
    out = [("the first records", 0), ("the last records", n - count)]
    end = rec_off(n - 1)
    for g in (4, 8, 16, 24, 32, 48):
        if g * GIB < end:
            out.append((f"records across stream offset {g} GiB", clamp(_lower_bound(n, rec_off, g * GIB) - count // 2)))
    tot_seq = seq_off(n - 1)
    for g in (4, 8, 12):
        if g * GIB < tot_seq:
            out.append((f"records across SEQ/QUAL arena offset {g} GiB", clamp(_lower_bound(n, seq_off, g * GIB) - count // 2)))
    runs = info.runs
    for g in (4, 8, 12):            # the dense column array: byte offset g GiB = column index g GiB / 32
        ci = g * GIB // 32
        for r in runs:
            cb, sp, ep = int(r["column_base"]), int(r["start_pos"]), int(r["end_pos"])
            if cb <= ci < cb + (ep - sp):
                p = sp + (ci - cb)
                out.append((f"columns across byte offset {g} GiB of the column array", clamp(_lower_bound(n, poskey, (int(r["ref_id"]) << 32) | p) - count // 2)))
    n_ref = int(cfg.n_ref)
    for c in sorted({1, n_ref // 2, n_ref - 1} - {0}):
        if c * per < n:
            out.append((f"contig boundary {c - 1}|{c}", clamp(c * per - count // 2)))
    if torch is not None and info.n_positions:      # the deepest column (cfg3: inside a pile of N-skip reads)
        cols = torch.as_tensor(_Dev(info.device_columns, (int(info.n_positions), 8), "<i4"), device=dev)
        best, best_i = -1, 0
        step = 1 << 26
        for o in range(0, int(info.n_positions), step):
            c = cols[o:o + step, 0]
            m = int(c.max().item())
            if m > best:
                best, best_i = m, o + int(c.argmax().item())
        for r in runs:
            cb, sp, ep = int(r["column_base"]), int(r["start_pos"]), int(r["end_pos"])
            if cb <= best_i < cb + (ep - sp):
                p = sp + (best_i - cb)
                out.append((f"the deepest column (coverage {best})", clamp(_lower_bound(n, poskey, (int(r["ref_id"]) << 32) | p) - count // 2)))
    seen, uniq = set(), []
    for lab, f in out:
        if f not in seen:
            seen.add(f); uniq.append((lab, f))
    k = 0
    while len(uniq) < want and k < 8 * want:          # fill up with ranges inside the contigs, spread over the stream
        k += 1
        f = clamp(n * k // (want + 1) + 12345)
        if f not in seen:
            seen.add(f); uniq.append((f"inside the stream at {k}/{want + 1}", f))
    return uniq


class _Dev:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2}


def check_range(ctx, cfg, first, count, entries=False, entry_positions=1500):
    """Raises AssertionError (with what differs) unless the device's products for records [first, first + count) are the
    oracle's.  Returns {"reads", "positions", "entries"} compared."""
    n = ctx.num_records()
    assert 0 <= first and first + count <= n
    per, span = reads_per_contig(cfg), max_span(cfg)
    contig_first = (first // per) * per
    stride = max(1, (int(cfg.ref_len) - span) // per)
    back = min(first - contig_first, span // stride + 64)
    hi = min(n, first + count + 1)
    while True:                                       # left halo: every earlier read that could reach the range's first position
        st = host_stream(cfg, first - back, hi - (first - back))
        soa = O.decode_soa(st)
        if back == first - contig_first or int(soa.pos[back]) - int(soa.pos[0]) >= span:
            break
        back = min(first - contig_first, 2 * back)
    assert len(soa.rec_off) == hi - (first - back)
    a, b = back, back + count                         # the range inside the host stream
    # ---- (a) rows, stream bytes, arenas
    d = ctx.decode_fetch_range(first, count, L.BGP_FETCH_ALL, stream=True)
    for f in FIELDS:
        assert np.array_equal(d.rec[f], getattr(soa, f)[a:b]), f"record table: {f}"
    ro = d.rec["rec_off"].astype(np.int64)
    assert np.array_equal(ro - ro[0], soa.rec_off[a:b].astype(np.int64) - int(soa.rec_off[a])), "record table: rec_off"
    s0 = int(soa.rec_off[a]); s1 = int(soa.rec_off[b]) if b < len(soa.rec_off) else len(st)
    assert d.data is not None and len(d.data) == s1 - s0 and d.data.tobytes() == st[s0:s1], "the records' stream bytes"
    so, co = d.seq_off.astype(np.int64), d.cigar_off.astype(np.int64)
    assert np.array_equal(so - so[0], soa.seq_off[a:b + 1].astype(np.int64) - int(soa.seq_off[a])), "seq_off"
    assert np.array_equal(co - co[0], soa.cigar_off[a:b + 1].astype(np.int64) - int(soa.cigar_off[a])), "cigar_off"
    if cfg.len_min == cfg.len_max:
        assert int(so[0]) == first * int(cfg.len_min), "seq_off is not the sum of l_seq in front of the range"
    assert np.array_equal(d.seq, soa.seq[int(soa.seq_off[a]):int(soa.seq_off[b])]), "SEQ arena slice"
    assert np.array_equal(d.qual, soa.qual[int(soa.seq_off[a]):int(soa.seq_off[b])]), "QUAL arena slice"
    assert np.array_equal(d.cigar, soa.cigar[int(soa.cigar_off[a]):int(soa.cigar_off[b])]), "CIGAR arena slice"
    # ---- (b) columns the range determines completely
    ora = O.pileup(st, skip_zero=False)
    r0 = int(soa.ref_id[a])
    r_last = int(soa.ref_id[b]) if b < len(soa.rec_off) else None
    n_pos = 0
    for r in sorted(set(int(x) for x in soa.ref_id[a:b])):
        lo = int(soa.pos[a]) if r == r0 else 0
        hi_p = int(soa.pos[b]) if r == r_last else None
        if hi_p is not None and hi_p <= lo:
            continue
        sel = (ora.ref == r) & (ora.pos >= lo) & ((ora.pos < hi_p) if hi_p is not None else True)
        wins = list(ctx.pileup_windows(L.BGP_MAKE_PILEUP, ref_id=r, start_from=lo, end_at=hi_p if hi_p is not None else (1 << 64) - 1,
                                       skip_zero_coverage=False, window_positions=1 << 20))
        dev_end = max([int(w.first_position) + len(w.columns) for w in wins], default=lo)
        ora_end = int(ora.pos[sel].max()) + 1 if sel.any() else lo
        top = max(dev_end, ora_end)
        exp = np.zeros(top - lo, L.COLUMN_DTYPE); got = np.zeros(top - lo, L.COLUMN_DTYPE)
        idx = ora.pos[sel].astype(np.int64) - lo
        exp["coverage"][idx] = ora.cov[sel]; exp["n_starting_here"][idx] = ora.nstart[sel]
        for j, f in enumerate(("a", "c", "g", "t", "other", "del")):
            exp[f][idx] = ora.counts[sel][:, j]
        for w in wins:
            assert w.ref_id == r and int(w.first_position) >= lo
            o = int(w.first_position) - lo
            got[o:o + len(w.columns)] = w.columns
        for f in L.COLUMN_DTYPE.names:
            bad = np.nonzero(got[f] != exp[f])[0]
            assert len(bad) == 0, f"columns of reference {r}: {f} differs at {len(bad)} positions, first {lo + int(bad[0])}: device {int(got[f][bad[0]])}, oracle {int(exp[f][bad[0]])}"
        n_pos += top - lo
    # ---- (c) per-read entries of a window inside the range (reference r0)
    n_ent = 0
    if entries:
        lo = int(soa.pos[a]); hi_p = int(soa.pos[b]) if r_last == r0 else int(soa.end_pos[a:b][soa.ref_id[a:b] == r0].max())
        wa = lo + max(0, (hi_p - lo - entry_positions) // 2); wb = min(hi_p, wa + entry_positions)
        ore = O.pileup(st, mode=O.MODE_MAKE_PILEUP, start_from=wa, end_at=wb, skip_zero=True, want_entries=True)
        assert ore.pileup_ref_id == r0 or ore.n_columns == 0
        k = 0
        for w in ctx.pileup_windows(L.BGP_MAKE_PILEUP, ref_id=r0, start_from=wa, end_at=wb, skip_zero_coverage=True, want_entries=True, window_positions=4096):
            for j in np.nonzero(w.columns["coverage"])[0]:
                e = w.entries[int(w.entry_off[j]):int(w.entry_off[j + 1])]
                oe = ore.entries[int(ore.entry_off[k]):int(ore.entry_off[k + 1])]
                assert int(ore.pos[k]) == int(w.first_position) + int(j), "entries: column positions"
                assert np.array_equal(e["read_ordinal"].astype(np.int64), oe["read_ordinal"].astype(np.int64) + (first - back)), f"entries: read_ordinal at {int(ore.pos[k])}"
                for f in ("cigar_op_index", "cigar_op_offset", "query_offset", "base", "qual"):
                    assert np.array_equal(e[f].astype(np.int64), oe[f].astype(np.int64)), f"entries: {f} at {int(ore.pos[k])}"
                n_ent += len(e); k += 1
        assert k == ore.n_columns and k > 0, "entries: number of columns"
    return {"reads": count, "positions": n_pos, "entries": n_ent}


def md_prefix_check(ctx, cfg, count=20_000):
    """reference_base with use_md_tag (PileupRangeUsingMdTag, pileup.d:522-654) on the resident result: the columns in front of
    record `count`'s position depend on records [0, count) alone -- add() sees a read when the sweep reaches its position
    (pileup.d:563-596) -- so the oracle's sweep over that prefix must give the device's bytes there.  (Only the stream's first
    reference: a later one begins under the chunk_end_position the one before left behind, pileup.d:583-595.)"""
    n = ctx.num_records()
    count = min(count, n - 1)
    st = host_stream(cfg, 0, count + 1)
    soa = O.decode_soa(st)
    r0 = int(soa.ref_id[0])
    same = np.nonzero(soa.ref_id != r0)[0]
    if len(same):
        count = int(same[0]) - 1
    hi_p = int(soa.pos[count])
    ora = O.pileup(host_stream(cfg, 0, count), use_md=True, skip_zero=True)
    sel = (ora.ref == r0) & (ora.pos < hi_p)
    poss, bases = [], []
    for w in ctx.pileup_windows(L.BGP_MAKE_PILEUP, use_md_tag=True, ref_id=r0, start_from=0, end_at=hi_p, skip_zero_coverage=True, window_positions=1 << 20):
        keep = w.columns["coverage"] > 0
        poss.append((int(w.first_position) + np.arange(len(w.columns), dtype=np.int64))[keep]); bases.append(w.reference_base[keep])
    poss = np.concatenate(poss) if poss else np.zeros(0, np.int64)
    bases = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
    assert np.array_equal(poss, ora.pos[sel].astype(np.int64)), "use_md_tag: the columns visited"
    bad = np.nonzero(bases != ora.refbase[sel])[0]
    assert len(bad) == 0, f"reference_base differs at {len(bad)} columns, first {int(poss[bad[0]])}: device {chr(bases[bad[0]])}, oracle {chr(ora.refbase[sel][bad[0]])}"
    return {"reads": count, "columns": int(len(poss))}


def md_contig_prefix_check(ctx, cfg, count=20_000):
    """The same at the start of EVERY contig of the stream (cfg3: records 0, 12.5 M, 25 M, ... -- reference_base bytes past 4 / 8 / 16 GiB
    of stream): bam[contig] piled up by itself (bgp_pileup_begin_ref: the contig's reads are the whole input, so its provider chain begins
    afresh, as PileupRangeUsingMdTag's does on a new range, pileup.d:573-579) against the oracle's sweep over the contig's first `count`
    records generated on the host.  Returns one entry per contig; raises AssertionError on the first difference."""
    n = ctx.num_records()
    per = reads_per_contig(cfg)
    out = []
    for c in range(int(cfg.n_ref)):
        s = c * per
        if s >= n:
            break
        cnt = min(count, n - s - 1, per - 1)
        if cnt < 2:
            continue
        st = host_stream(cfg, s, cnt + 1)
        soa = O.decode_soa(st)
        rid = int(soa.ref_id[0])
        assert (soa.ref_id == rid).all()
        hi_p = int(soa.pos[cnt])
        ora = O.pileup(host_stream(cfg, s, cnt), use_md=True, skip_zero=True)
        sel = (ora.ref == rid) & (ora.pos < hi_p)
        poss, bases = [], []
        for w in ctx.pileup_windows(L.BGP_MAKE_PILEUP, use_md_tag=True, ref_id=rid, start_from=0, end_at=hi_p, skip_zero_coverage=True, window_positions=1 << 20):
            keep = w.columns["coverage"] > 0
            poss.append((int(w.first_position) + np.arange(len(w.columns), dtype=np.int64))[keep]); bases.append(w.reference_base[keep])
        poss = np.concatenate(poss) if poss else np.zeros(0, np.int64)
        bases = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
        assert np.array_equal(poss, ora.pos[sel].astype(np.int64)), f"use_md_tag, contig {rid}: the columns visited"
        bad = np.nonzero(bases != ora.refbase[sel])[0]
        assert len(bad) == 0, f"contig {rid}: reference_base differs at {len(bad)} columns, first {int(poss[bad[0]])}: device {chr(bases[bad[0]])}, oracle {chr(ora.refbase[sel][bad[0]])}"
        out.append({"contig": rid, "first_record": int(s), "reads": int(cnt), "columns": int(len(poss))})
    return out


def synth_reference_chars(seed, contig, p0, n):
    """The generator's reference (biod_amd/csrc/synth_record.h: ref2), positions [p0, p0 + n) of contig `contig`, as chars: a hash per block of 32
    positions, two bits a base."""
    M = np.uint64

    def mix64(z):
        z = z + M(0x9E3779B97F4A7C15)
        z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
        return z ^ (z >> M(31))
    if n <= 0:
        return np.zeros(0, np.uint8)
    b0, b1 = p0 >> 5, (p0 + n - 1) >> 5
    with np.errstate(over="ignore"):
        h = mix64(M(seed) ^ M(0x5EEDBA5E) ^ (M(contig) << M(36)) ^ np.arange(b0, b1 + 1, dtype=np.uint64))
    chars = _REF_LUT[h.astype("<u8").view(np.uint8)].reshape(-1)      # (a byte of the hash is four bases, lowest bits first)
    return chars[(p0 & 31):(p0 & 31) + n]


_REF_LUT = np.frombuffer(b"ACGT", np.uint8)[(np.arange(256)[:, None] >> (2 * np.arange(4))[None, :]) & 3]


def md_generator_reference_check(ctx, cfg, window_positions=1 << 22):
    """A property of the generator, good at any size and wherever in the stream: its MD tags are written against ONE hash-defined reference,
    so whichever read provides a column, reference_base (PileupRangeUsingMdTag, pileup.d:522-654) must be that reference's base -- on streams
    without N skips (dna(read) leaves skipped bases out, md/reconstruct.d:199: behind an N a chunk drifts off the reference, in BioD and
    here alike; cfg3 has them, cfg2 and cfg5 do not).  Every covered column of the whole piled-up stream; uncovered ones inside a run hold
    'N' (an empty chunk, pileup.d:620-622)."""
    n_cols = n_bad = n_unc_bad = 0
    first_bad = None
    for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=True, skip_zero_coverage=False, window_positions=window_positions, copy=False):
        keep = w.columns["coverage"] > 0
        exp = synth_reference_chars(int(cfg.seed), int(w.ref_id), int(w.first_position), len(w.columns))
        bad = np.nonzero((w.reference_base != exp) & keep)[0]
        n_cols += int(keep.sum()); n_bad += len(bad)
        n_unc_bad += int((w.reference_base[~keep] != ord("N")).sum())
        if len(bad) and first_bad is None:
            first_bad = (int(w.ref_id), int(w.first_position) + int(bad[0]), chr(w.reference_base[bad[0]]), chr(exp[bad[0]]))
    assert n_bad == 0 and n_unc_bad == 0, f"reference_base differs from the generator's reference at {n_bad} covered columns (first {first_bad}); {n_unc_bad} uncovered columns are not 'N'"
    return {"covered_columns": n_cols, "equal": True}


def run(ctx, cfg, info, count, torch=None, dev=None, want=16, entries_on=(0, 5)):
    """All planned ranges; returns the summary bench.py prints as `parity_interior`."""
    plan = plan_ranges(ctx, cfg, info, count, torch, dev, want)
    res = {"ranges": 0, "reads": 0, "positions": 0, "entries": 0, "all_equal": True, "checked": [], "failed": []}
    for k, (label, first) in enumerate(plan):
        try:
            r = check_range(ctx, cfg, first, count, entries=k in entries_on)
            res["ranges"] += 1; res["reads"] += r["reads"]; res["positions"] += r["positions"]; res["entries"] += r["entries"]
            res["checked"].append(f"{label} [{first}, +{count})" + (" +entries" if r["entries"] else ""))
        except AssertionError as e:
            res["all_equal"] = False
            res["failed"].append(f"{label} [{first}, +{count}): {e}")
    return res
