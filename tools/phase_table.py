#!/usr/bin/env python3
"""Phase table of ONE replayed CNE+SUE training step (tools/tape_timeline.py's per-call events folded into the phases of the step's
dependent chain), for same-box A/Bs of an environment switch:

    NNR_BX3=0 python tools/phase_table.py --json gpurun_out/p0.json ;  NNR_BX3=1 python tools/phase_table.py --json gpurun_out/p1.json
    python tools/phase_table.py --diff gpurun_out/p0.json gpurun_out/p1.json

Phases (boundaries = starts of the named calls on the step's timeline):
  projection        first call .. lstm_fwd             (plans, embedding gather, input projection GEMMs, token sort)
  recurrence_fwd    lstm_fwd
  post_recurrence   end of lstm_fwd .. sue_x0_fwd      (gates, attention projections, pools, fusion rows)
  sue               sue_x0_fwd .. first dyn pool_bwd   (user encoder forward + backward, click loss)
  cne_bwd_prologue  first dyn pool_bwd .. lstm_bwd     (pool / attention / gate backward)
  recurrence_bwd    first lstm_bwd start .. last lstm_bwd end
  tail              .. sumsq                            (dX, dW_ih, dW_hh, embedding-row scatter)
  optimizer         sumsq .. end
Per phase: wall span, the sum of the calls' own durations by family inside it, and how long exactly one / >= 2 calls were in flight."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ['projection', 'recurrence_fwd', 'post_recurrence', 'sue', 'cne_bwd_prologue', 'recurrence_bwd', 'tail', 'optimizer']


def boundaries(rows):
    """rows: [(start_ms, dur_ms, stream, family, tag)] sorted by start."""
    def first(pred, after=0.0):
        for s, d, st, fam, tag in rows:
            if s >= after and pred(fam, tag):
                return s, s + d
        return None
    f = first(lambda fam, tag: fam == 'lstm_fwd')
    x0 = first(lambda fam, tag: fam == 'sue_x0_fwd')
    pb = first(lambda fam, tag: fam == 'pool_bwd' and 'dyn' in tag, x0[0] if x0 else 0.0)
    lb = [(s, s + d) for s, d, st, fam, tag in rows if fam == 'lstm_bwd']
    sq = first(lambda fam, tag: fam == 'sumsq')
    end = max(s + d for s, d, *_ in rows)
    if not (f and x0 and pb and lb and sq):
        return None
    cuts = [0.0, f[0], f[1], x0[0], pb[0], min(a for a, _ in lb), max(b for _, b in lb), sq[0], end]
    return cuts


def fold(rows):
    rows = sorted(rows)
    cuts = boundaries(rows)
    if cuts is None:
        return None
    out = {}
    for i, name in enumerate(PHASES):
        lo, hi = cuts[i], cuts[i + 1]
        fam = {}
        ev = []
        for s, d, st, f, tag in rows:
            a, b = max(s, lo), min(s + d, hi)
            if b > a:
                key = f + ((' ' + tag.split(' dyn')[0]) if f.startswith('gemm') else '')
                fam[key] = fam.get(key, 0.0) + (b - a)
                ev += [(a, 1), (b, -1)]
        ev.sort()
        busy = [0.0, 0.0, 0.0]
        lvl, last = 0, lo
        for t, k in ev:
            busy[min(lvl, 2)] += t - last
            lvl += k
            last = t
        busy[0] += hi - last
        out[name] = {'span_ms': round(hi - lo, 4), 'idle_ms': round(busy[0], 4), 'one_ms': round(busy[1], 4), 'multi_ms': round(busy[2], 4),
                     'families_ms': {k: round(v, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8]}}
    return out


def measure(a):
    import time
    import numpy as np
    import torch
    from nnr_amd import tape as T
    from nnr_amd.config import make_config
    from nnr_amd.model import Model
    from nnr_amd.synth import SynthSpec, SynthCorpus, to_torch
    from nnr_amd.trainer import Trainer
    T.TAG_ALL[0] = True
    cfg = make_config(['--news_encoder=CNE', '--user_encoder=SUE', '--dataset=200k', '--batch_size=%d' % a.batch_size],
                      corpus_sizes=dict(vocabulary_size=a.vocabulary_size))
    torch.manual_seed(0)
    table = torch.randn(cfg.vocabulary_size, cfg.word_embedding_dim) * 0.3
    table[0] = 0
    model = Model(cfg, table)
    model.initialize()
    tr = Trainer(model.cuda().train(), cfg)
    corpus = SynthCorpus(SynthSpec(vocabulary_size=cfg.vocabulary_size))
    rng = np.random.default_rng(100)
    batches = [to_torch(corpus.batch(a.batch_size, rng), 'cuda') for _ in range(4)]
    for i in range(6):
        tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
    plain = []
    for rep in range(a.reps):
        t0 = time.perf_counter()
        for i in range(20):
            tr.train_step(batches[0])
        torch.cuda.synchronize()
        plain.append((time.perf_counter() - t0) / 20 * 1000)
    tables = []
    for rep in range(a.timed):
        tr.timing = True
        tr.train_step(batches[0])
        tr.timing = False
        torch.cuda.synchronize()
    tape = next(iter(tr.tapes.values()))
    for rep in range(a.timed):
        t = fold(tape.timeline(rep))
        if t is not None:
            tables.append(t)
    env = {k: v for k, v in os.environ.items() if k.startswith('NNR_')}
    return {'env': env, 'batch_size': a.batch_size, 'untimed_ms': [round(x, 4) for x in plain], 'tables': tables}


def median_table(tables):
    import statistics
    out = {}
    for name in PHASES:
        out[name] = {k: round(statistics.median(t[name][k] for t in tables), 4) for k in ('span_ms', 'idle_ms', 'one_ms', 'multi_ms')}
        fams = {}
        for t in tables:
            for k, v in t[name]['families_ms'].items():
                fams.setdefault(k, []).append(v)
        out[name]['families_ms'] = {k: round(statistics.median(v), 4) for k, v in sorted(fams.items(), key=lambda kv: -statistics.median(kv[1]))[:6]}
    return out


def diff(pa, pb):
    A, B = json.load(open(pa)), json.load(open(pb))
    ta, tb = median_table(A['tables']), median_table(B['tables'])
    print('A: %s  untimed %s ms' % (A['env'], A['untimed_ms']))
    print('B: %s  untimed %s ms' % (B['env'], B['untimed_ms']))
    print('| phase | A span ms | B span ms | B - A | A busiest families (ms of call time inside the phase) | B busiest families |')
    print('|---|---|---|---|---|---|')
    tot = 0.0
    for name in PHASES:
        d = tb[name]['span_ms'] - ta[name]['span_ms']
        tot += d
        fa = ', '.join('%s %.2f' % kv for kv in list(ta[name]['families_ms'].items())[:3])
        fb = ', '.join('%s %.2f' % kv for kv in list(tb[name]['families_ms'].items())[:3])
        print('| %s | %.3f | %.3f | %+.3f | %s | %s |' % (name, ta[name]['span_ms'], tb[name]['span_ms'], d, fa, fb))
    print('| sum | %.3f | %.3f | %+.3f | | |' % (sum(ta[n]['span_ms'] for n in PHASES), sum(tb[n]['span_ms'] for n in PHASES), tot))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch_size', type=int, default=64)
    ap.add_argument('--vocabulary_size', type=int, default=60000)
    ap.add_argument('--reps', type=int, default=3, help='un-instrumented 20-step windows')
    ap.add_argument('--timed', type=int, default=5, help='replays under per-call events (median table)')
    ap.add_argument('--json', default=None)
    ap.add_argument('--diff', nargs=2, default=None)
    a = ap.parse_args()
    if a.diff:
        diff(*a.diff)
        sys.exit(0)
    res = measure(a)
    if a.json:
        json.dump(res, open(a.json, 'w'))
    mt = median_table(res['tables']) if res['tables'] else {}
    print('untimed 20-step windows: %s ms; env %s' % (res['untimed_ms'], res['env']))
    for name in PHASES:
        if name in mt:
            print('%-18s span %7.3f  idle %6.3f  one %6.3f  multi %6.3f   %s' % (name, mt[name]['span_ms'], mt[name]['idle_ms'], mt[name]['one_ms'], mt[name]['multi_ms'],
                                                                                ', '.join('%s %.2f' % kv for kv in list(mt[name]['families_ms'].items())[:4])))
