#!/bin/bash
# profiles/resource_usage.sh > profiles/TAG_resource_usage.txt — registers, LDS, scratch and occupancy of every kernel
# (hipcc -Rpass-analysis=kernel-resource-usage, names demangled); needs no GPU.
cd "$(dirname "$0")/../hardware-acceleration-of-lidar-slam_amd/csrc"
for f in score_kernels edt_kernels pf_kernels paged_kernels mapper_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
      -Rpass-analysis=kernel-resource-usage -c -o /dev/null $f.hip 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize"
done | python3 -c "
import sys, re, subprocess
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for k in ('TotalSGPRs', 'VGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]'):
        m = re.search(re.escape(k) + r': (\d+)', ln)
        if m and cur is not None and k not in cur:
            cur[k] = m.group(1)
print('# per-kernel resources, gfx950, hipcc -O3 -ffp-contract=off (-Rpass-analysis=kernel-resource-usage)')
print('%-5s %-5s %-7s %-4s %-7s  %s' % ('VGPR', 'SGPR', 'scratch', 'occ', 'LDS', 'kernel'))
seen = set()
for r in rows:
    n = re.sub(r'\(slam::.*|\((float|int|unsigned|long|slam).*', '', r['name']).replace('slam::(anonymous namespace)::', '').replace('void ', '')
    if n in seen:
        continue
    seen.add(n)
    print('%-5s %-5s %-7s %-4s %-7s  %s' % (r.get('VGPRs', '?'), r.get('TotalSGPRs', '?'), r.get('ScratchSize [bytes/lane]', '?'),
                                          r.get('Occupancy [waves/SIMD]', '?'), r.get('LDS Size [bytes/block]', '?'), n))
"
