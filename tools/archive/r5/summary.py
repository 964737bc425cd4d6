#!/usr/bin/env python3
"""Writes profiles/r05_summary.md from the committed round-5 artefacts (profiles/r05f_*, pmc_*.json).  No GPU."""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, 'profiles')
steps = 16


def short(n, w=58):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:w]


def jl(f):
    return json.load(open(os.path.join(P, 'r05f_%s.json' % f)))


rows = list(csv.DictReader(open(os.path.join(P, 'r05f_one_stream_kernel_stats.csv'))))
tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6
solo = ['| `%s` | %.1f | %.1f | %.3f |' % (short(r['Name']), int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e6) for r in rows[:24]]
rows2 = list(csv.DictReader(open(os.path.join(P, 'r05f_kernel_stats.csv'))))
instep = ['| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])) for r in rows2[:14]]
b, bt, pb, mh, so = jl('bench'), jl('bench_with_traffic'), jl('bench_prebuilt'), jl('bench_mhsa'), jl('soak')


def row(n, d):
    return '| %s | %.1f | %.3f | %s | %.1f | %.3f |' % (n, d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step'), d['roofline']['step']['gflop'],
                                                      d['roofline']['step']['frac'])


out = ['# Round 5 -- final build (`r05f_*`), 1x MI355X\n']
out.append('Build id %s (`profiles/pmc_traffic.json`, `profiles/pmc_mfma_busy.json` carry the same id).  All lines: `python3 bench.py ...` on one gpurun box, back to back '
           '(`tools/r5/collect.sh r05f`); this file: `tools/r5/summary.py`.\n' % json.load(open(os.path.join(P, 'pmc_traffic.json')))['build_id']['src_sha256'])
out.append('| run | impressions/s | ms/step (window) | ms/step (sustained) | GFLOP/step | step frac of fp32 MFMA peak |\n|---|---|---|---|---|---|')
out.append(row('default (first GPU process of the box; + cpu_baseline leg)', b))
out.append(row('default again, PMC traffic quoted', bt))
out.append(row('`--prebuilt` (rounds 1-3 timed region)', pb))
for n in ['b8', 'b16', 'b32', 'b128']:
    out.append(row('`--batch_size %s`' % n[1:], jl('bench_' + n)))
out.append(row('`--config mhsa` (BASELINE configs[1]; news encoder over packed token rows)', mh))
out.append(row('`--config mhsa`, `NNR_MHSA_PACKED=0` (rounds 1-4: all padded rows)', jl('bench_mhsa_dense_rows')))
out.append('')
out.append('`secondary` legs of the default line (fresh model + trainer each, 10 replayed steps after 5 warm-up steps):\n')
out.append('| leg | config | ms/step | impressions/s (this GPU\'s shard) | GFLOP/step | step frac |\n|---|---|---|---|---|---|')
for k, v in bt['secondary'].items():
    out.append('| %s | %s | %.3f | %.1f | %.1f | %.3f |' % (k, v['config'], v['ms_per_step'], v['value'], v['step']['gflop'], v['step']['frac']))
out.append('')
r = bt['roofline']
out.append('Dominant family of the default line: `%s` (`%s`), %d launches sampled, %.1f us average in-step, %.1f TFLOP/s = %.3f of 157.3; HBM counter traffic %.1f MB per launch vs %.1f MB of '
           'operands (x%.2f).  The same launches with every stream collapsed into one: %.1f TFLOP/s = %.3f (`roofline.isolated`).  CPU baseline: %.2f impressions/s (%s).\n' % (
               r['family'], r['kernel'], r['launches'], r['avg_launch_us'], r['achieved'], r['frac'], r['traffic'] / 1e6, r['algorithmic_bytes_per_launch'] / 1e6,
               r['traffic_over_algorithmic'], r['isolated']['achieved'], r['isolated']['frac'], bt['cpu_baseline']['value'], bt['cpu_baseline']['sample'][:120]))
out.append('`roofline.hbm` of the same line (algorithmic GB/s in-step, fraction of 8 TB/s, counter / algorithmic bytes):\n')
out.append('| kernel | launches | us | GB/s | frac | counter/algorithmic |\n|---|---|---|---|---|---|')
for k, v in sorted(r['hbm'].items(), key=lambda kv: -kv[1]['achieved']):
    out.append('| %s | %d | %.1f | %.0f | %.3f | %s |' % (k, v['launches'], v['avg_launch_us'], v['achieved'], v['frac'], v['traffic_over_algorithmic']))
out.append('')
m = mh['roofline']['mhsa']
out.append('`roofline.mhsa` (`--config mhsa`): ' + '; '.join('%s %.0f us, %.1f TFLOP/s of MFMA work (%.3f), %.0f GB/s (%.3f), matrix-pipe busy counter %.3f' % (
    k, v['avg_launch_us'], v['mfma_tflops'], v['mfma_utilisation'], v['hbm_gb_s'], v['hbm_frac'], v['mfma_busy_counter']) for k, v in m.items()) + '.\n')
out.append('Soak (`tools/replay_soak.py --steps 1500`): %.3f ms/step including 16 loss read-backs, paths %s, recurrence exchange time-outs %d, Adam steps skipped %d, parameters finite %s.\n' % (
    [v for k, v in so.items() if k.startswith('ms_per_step')][0], so['paths'], so['recurrence_exchange_timeouts'], so['adam_steps_skipped'], so['parameters_finite']))
out.append('## Solo kernel durations (`NNR_ONE_STREAM=1`, rocprofv3 --kernel-trace --stats, `r05f_one_stream_kernel_stats.csv`): %.2f ms of kernel time per step\n' % tot)
out.append('| kernel | launches / step | average us | ms / step |\n|---|---|---|---|')
out += solo
out.append('\n## In-step kernel table (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 12 --warmup 4`, `r05f_kernel_stats.csv`; four-five streams: durations overlap)\n')
out.append('| kernel | calls | average us | % of summed kernel time |\n|---|---|---|---|')
out += instep
dom = [x for x in rows2 if r['kernel'] in x['Name']]
one = [x for x in rows if r['kernel'] in x['Name']]
out.append('\nA/B records of the round: `profiles/r05_ab.txt`.  Timelines of a replayed step: `profiles/r05_timeline_b64.txt`, `profiles/r05_timeline_b16.txt`, `profiles/r05_timeline_b8.txt`.  bf16x3 micro-benchmark: `profiles/r05_bf16x3.txt`.\n')
out.append('Live vs rocprofv3 duration of the dominant family: the bench line\'s HIP-event pair brackets the launch ON ITS STREAM (previous command\'s end -> kernel\'s end), so it includes the time the '
           'launch\'s workgroups queue for CUs held by the other streams\' kernels (%.1f us between first wave and last wave in the rocprofv3 trace of the same command vs %.0f us between the events; with '
           'one stream %.1f us in the trace).  `roofline.frac` uses the event figure (the pessimistic one); `roofline.isolated` is the one-stream figure.' % (
               float(dom[0]['AverageNs']) / 1e3, r['avg_launch_us'], float(one[0]['AverageNs']) / 1e3))
for b, n in (('busy', 'per-GPU batch 8 beside 48 resident 512-thread workgroups (nnr_dp_busy)'),):
    f = os.path.join(P, 'r05_soak_%s.json' % b)
    if os.path.exists(f):
        d = json.load(open(f))
        ms = [v for k, v in d.items() if k.startswith('ms_per_step')][0]
        out.append('\nLong soak at %s (`%s`): %d replayed steps, %.3f ms/step including the loss read-backs, %d recurrence exchange time-outs, %d skipped optimizer steps, parameters finite %s.'
                   % (n, os.path.basename(f), d['steps'], ms, d['recurrence_exchange_timeouts'], d['adam_steps_skipped'], d['parameters_finite']))
open(os.path.join(P, 'r05_summary.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out)[:2600])
