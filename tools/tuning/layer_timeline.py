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
a, b = att[-2], att[-1]                       # layers 11 -> 12 of the last replay
t0 = int(rows[a]['Start_Timestamp'])
print('# One encoder layer of the DEFAULT-route BERT-base forward ([8,128], fixed ranges) inside a hipGraph replay:')
print('# rocprofv3 --kernel-trace over tools/tuning/bert_default_prof.py; start offset / duration of every kernel between two')
print('# consecutive attention cores (layers 11 -> 12 of the last replay; tools/tuning/layer_timeline.py).  The kernels run')
print('# back to back: the forward is the sum of 7 latency-bound launches per layer, not launch gaps.')
if len(sys.argv) > 2:                          # any other model: no per-kernel legend
    WHAT = []
    print(f'# ({sys.argv[2]})')
for k, r in enumerate(rows[a:b + 1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0]
    print(f"{(s - t0) / 1e3:8.1f} us  + {(e - s) / 1e3:5.1f} us  {name:52s} {WHAT[k] if k < len(WHAT) else ''}")
