#!/bin/bash
# scripts/c5_laps.sh -- the product CLI on the configs[4] twin (64 genomes, -a -s 1500) with AGC_AMD_LAPS=1: host laps summed per stage
# (where the time of a small, adaptive collection goes: 64 registrations of one file each, every one a chain of short device calls)
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
from tests import collections as C
files = C.build('syn_c5_twin', '/tmp/c5in')
open('/tmp/c5files.json','w').write(json.dumps(files))
print(len(files), C.CONFIGS['syn_c5_twin'][0])
PY
FILES=$(python -c "import json; print(' '.join(json.load(open('/tmp/c5files.json'))))")
agc_amd/bin/agc_amd create -a -s 1500 -t 16 -o /tmp/o1.agc $FILES > /dev/null 2>&1
AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 agc_amd/bin/agc_amd create -a -s 1500 -t 16 -o /tmp/o2.agc $FILES 2> gpurun_out/c5_laps.txt
python - <<'PY'
import re, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for line in open('gpurun_out/c5_laps.txt'):
    m = re.match(r'\s*(?:lz_encode_end |entropy |close )?lap (.*?) ([0-9.e+-]+) ms', line)
    if m:
        tot[m.group(1)] += float(m.group(2)); cnt[m.group(1)] += 1
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:30]:
    print(f'{k:45s} {v:9.2f} ms  x{cnt[k]}')
print('sum', sum(tot.values()))
PY
