#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table for one .hip source (clang's kernel-resource-usage remarks).

    python tools/kernel_resources.py transformer-quantization_amd/csrc/tq_mse_ordered.hip [filter]
"""
import re
import subprocess
import sys

sys.path.insert(0, 'transformer-quantization_amd')
from build import flags_for  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc'] + flags_for(src) + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name)
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('SGPRs','?'):>4} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
