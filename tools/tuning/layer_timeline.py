"""Kernel timeline of ONE encoder layer of the default-route BERT-base forward inside a hipGraph replay, from a
rocprofv3 kernel trace of tools/tuning/bert_default_prof.py:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/tuning/bert_default_prof.py
    python tools/tuning/layer_timeline.py OUT > profiles/rNN/bert_default_route_layer_timeline.txt"""
import csv, glob, sys

WHAT = ['attention core (QK^T, quantizers, softmax, PV)', 'attention-output Linear (pre-quantizer output)',
        'residual + LayerNorm tail', 'FFN1 Linear + GELU + quantizer (index only)', 'FFN2 Linear (K = 3072)',
        'residual + LayerNorm tail', 'grouped Q|K|V Linear of the next layer (index only)', '']
t = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
att = [i for i, r in enumerate(rows) if 'attention_i8_k' in r['Kernel_Name']]
# the layer of the last replays whose kernels ran without a profiler-induced gap: the closest pair of consecutive cores
cand = list(zip(att[-12:-1], att[-11:]))
a, b = min(cand, key=lambda ab: int(rows[ab[1]]['Start_Timestamp']) - int(rows[ab[0]]['Start_Timestamp']))
t0 = int(rows[a]['Start_Timestamp'])
other = len(sys.argv) > 2
print(f"# One encoder layer of the DEFAULT-route {'MobileBERT W4A4' if other else 'BERT-base'} forward ([8,128], fixed ranges) inside a hipGraph replay:")
print(f"# rocprofv3 --kernel-trace over tools/tuning/{'mb' if other else 'bert'}_default_prof.py; start offset / duration of every kernel between two")
print('# consecutive attention cores (the tightest such pair of the last replay: rocprofv3 itself opens gaps of tens of us after')
print('# some kernels, which the un-profiled forward time does not contain; tools/tuning/layer_timeline.py).  The kernels run')
print('# back to back: the forward is the sum of its latency-bound launches, not launch gaps.')
if other:                                      # MobileBERT: attention core | attention output + tail | the four feed-forward
    WHAT = ['attention core', 'attention-output Linear + residual NoNorm tail', 'the FOUR feed-forward blocks (one launch)',
            'output bottleneck 128 -> 512 + residual NoNorm tail',
            'input bottlenecks + value Linear of the next layer (one grouped launch)',
            'query | key Linears (one grouped launch, index only)', '']
for k, r in enumerate(rows[a:b + 1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0]
    print(f"{(s - t0) / 1e3:8.1f} us  + {(e - s) / 1e3:5.1f} us  {name:52s} {WHAT[k] if k < len(WHAT) else ''}")
