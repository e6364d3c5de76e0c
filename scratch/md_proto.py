"""Prototype of the EVENT-LEVEL restatement of PileupRangeUsingMdTag (pileup.d:522-654): instead of
stepping every column, jump between events (read admissions, chunk fetches, coverage gaps, run ends)
and emit provider intervals.  Validated here against the oracle; then ported to HIP."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
from biod_amd import bamio

INF = 1 << 62


def event_level(reads, skip_zero):
    """reads: covered reads in file order: dict(ref,pos,end,dna(str)).
    Returns {(ref, col): base} for every column the sweep visits (pileupColumns semantics) or 'ERR'."""
    n = len(reads)
    # machine state; persists across runs (stale on purpose, like the D object's fields)
    P = None; L = 0; ci = 0; CE = 0; has_next = False; NX = None; curr_ref = -1
    i = 0
    intervals = []   # [ref, col_start, provider or None, ci_start]   in emission order
    visited = []     # (ref, first_col, last_col) ranges of visited columns

    def add(R, had_zero):                                   # pileup.d:563-596
        nonlocal has_next, NX, curr_ref
        if R['ref'] != curr_ref:
            curr_ref = R['ref']; has_next = True; NX = R; return
        if R['pos'] > CE and not had_zero:
            return
        if R['end'] > CE:
            if not has_next:
                has_next = True; NX = R
            elif R['end'] > NX['end']:
                NX = R

    while i < n:
        ref = reads[i]['ref']
        maxend = 0

        def init_new_reference(x):                          # pileup.d:598-611 (+ base :399-424)
            nonlocal i, maxend, P, L, ci, CE, has_next
            while i < n and reads[i]['ref'] == ref and reads[i]['pos'] == x:
                add(reads[i], True); maxend = max(maxend, reads[i]['end']); i += 1
            if has_next:
                P = NX; L = len(P['dna']); ci = 0; CE = P['end']; has_next = False
                if L == 0:
                    return False
                intervals.append([ref, x, P, 0]); ci = 1
            else:
                intervals.append([ref, x, None, 0])         # 'N' at x ...
                if P is not None and ci < L:
                    intervals.append([ref, x + 1, P, ci])   # ... and the old chunk simply goes on
            return True

        x = reads[i]['pos']
        if not init_new_reference(x):
            return 'ERR'
        y = x; seg_from = x
        while True:
            nxt_pos = reads[i]['pos'] if (i < n and reads[i]['ref'] == ref) else INF
            f = y + max(1, L - ci) if has_next else INF      # first visited column whose check sees an empty chunk
            E = max(y + 1, maxend)                           # first column >= y+1 without a survivor
            E_evt = E if (skip_zero or nxt_pos == INF) else INF
            if E_evt <= f and E_evt < nxt_pos:
                # coverage ends at E with nothing admitted there: jump (skip_zero) or end of run / of everything
                ci = min(L, ci + (E - y))                    # every popFront consumed a char first (pileup.d:615-622)
                visited.append((ref, seg_from, E - 1))
                if nxt_pos == INF:
                    break
                if not init_new_reference(nxt_pos):
                    return 'ERR'
                y = nxt_pos; seg_from = nxt_pos
                continue
            t = min(nxt_pos, f)
            ci = min(L, ci + (t - y))
            prev_zero = not (maxend > t - 1)                 # _prev_coverage == 0, pileup.d:565,625
            while i < n and reads[i]['ref'] == ref and reads[i]['pos'] == t:
                add(reads[i], prev_zero); maxend = max(maxend, reads[i]['end']); i += 1
            if ci >= L and has_next:                         # pileup.d:633-652
                P = NX; L = len(P['dna']); CE = P['end']; has_next = False
                ci = min(t - P['pos'], L)
                if ci >= L:
                    return 'ERR'
                intervals.append([ref, t, P, ci]); ci += 1
            y = t
    # ---- materialise
    res = {}
    for ref, a, b in visited:
        for col in range(a, b + 1):
            res[(ref, col)] = 'N'
    for k, (ref, col, prov, c0) in enumerate(intervals):
        end = INF
        if k + 1 < len(intervals) and intervals[k + 1][0] == ref:
            end = intervals[k + 1][1]
        if prov is None:
            continue
        d = prov['dna']; c = col; j = c0
        while c < end and j < len(d) and (ref, c) in res:
            res[(ref, c)] = d[j]; j += 1; c += 1
    return res


def oracle_reads(st):
    offs = O.frame(st)
    reads = []
    for k, off in enumerate(offs):
        f = O.fields_at(st, int(off))
        if f.bases_covered > 0:
            try:
                d = O.dna_read(st, int(off))
            except O.OracleError:
                d = ""
            reads.append(dict(ref=f.ref_id, pos=f.pos, end=f.end_pos, dna=d, idx=k))
    return reads


def compare(st, skip_zero, verbose=False):
    try:
        ora = O.pileup(st, O.MODE_PILEUP_COLUMNS, use_md=True, skip_zero=skip_zero)
        exp = {(int(r), int(p)): chr(b) for r, p, b in zip(ora.ref, ora.pos, ora.refbase)}
    except O.OracleError as e:
        exp = 'ERR' if e.code == O.ORC_EMD else ('OTHER', e.code)
    got = event_level(oracle_reads(st), skip_zero)
    if exp == 'ERR' or got == 'ERR':
        return exp == got, (exp if isinstance(exp, str) else len(exp), got if isinstance(got, str) else len(got))
    if isinstance(exp, tuple):
        return True, exp
    bad = [(k, exp[k], got.get(k)) for k in exp if got.get(k) != exp[k]]
    extra = [k for k in got if k not in exp]
    return (not bad and not extra), (bad[:5], extra[:5], len(exp))


def random_stream(rng, n_reads, p_gap=0.02, p_nomd=0.0, p_n=0.0, p_badmd=0.0, nref=2):
    ref_seq = {}
    recs = []; pos = 0; ref = 0
    for i in range(n_reads):
        if rng.random() < 0.01 and ref < nref - 1:
            ref += 1; pos = int(rng.integers(0, 30))
        pos += int(rng.integers(0, 8))
        if rng.random() < p_gap:
            pos += int(rng.integers(30, 200))
        ops = []
        kind = rng.random()
        ln = int(rng.integers(20, 60))
        if kind < 0.6: ops = [(ln, 'M')]
        elif kind < 0.7: ops = [(int(rng.integers(1, 8)), 'S'), (ln, 'M')]
        elif kind < 0.8: ops = [(ln // 2, 'M'), (int(rng.integers(1, 4)), 'I'), (ln - ln // 2, 'M')]
        elif kind < 0.9: ops = [(ln // 2, 'M'), (int(rng.integers(1, 6)), 'D'), (ln - ln // 2, 'M')]
        else: ops = [(ln // 2, '='), (ln - ln // 2, 'X')]
        if rng.random() < p_n:
            ops = [(ln // 2, 'M'), (int(rng.integers(5, 40)), 'N'), (ln - ln // 2, 'M')]
        qlen = sum(l for l, o in ops if o in 'MIS=X')
        seq = ''.join('ACGT'[int(x)] for x in rng.integers(0, 4, qlen))
        # MD from the read itself with a few mismatches/deletion strings
        md = ''; run = 0
        for l, o in ops:
            if o in 'M=X':
                for _ in range(l):
                    if rng.random() < 0.05:
                        md += str(run) + 'ACGT'[int(rng.integers(0, 4))]; run = 0
                    else:
                        run += 1
            elif o == 'D':
                md += str(run) + '^' + ''.join('ACGT'[int(x)] for x in rng.integers(0, 4, l)); run = 0
        md += str(run)
        if rng.random() < p_badmd:
            md = md[: max(1, len(md) // 2)]
        tags = b'' if rng.random() < p_nomd else bamio.tag_z('MD', md)
        recs.append(bamio.pack_record(f'r{i}', seq, ops, ref_id=ref, pos=pos, tags=tags))
    return b''.join(recs)


if __name__ == '__main__':
    rng = np.random.default_rng(5)
    total = 0; fails = 0
    for trial in range(300):
        cfgs = [dict(), dict(p_gap=0.1), dict(p_n=0.1), dict(p_nomd=0.05), dict(p_badmd=0.05), dict(p_gap=0.05, p_n=0.05, p_nomd=0.02, p_badmd=0.02, nref=3)]
        kw = cfgs[trial % len(cfgs)]
        st = random_stream(rng, int(rng.integers(5, 120)), **kw)
        for skip in (True, False):
            ok, info = compare(st, skip)
            total += 1
            if not ok:
                fails += 1
                if fails <= 5:
                    print('MISMATCH trial', trial, kw, 'skip', skip, info)
    print('total', total, 'fails', fails)
