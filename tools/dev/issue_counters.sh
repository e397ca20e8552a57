#!/bin/bash
# Developer tool (GPU box): issue counters of the largest kernels, one line each (profiles/collect_valu.sh).
cd $GRAFT_REPO_ROOT
bash profiles/collect_valu.sh r5q_v > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5q_v/pmc_valu.json'))
for k,v in d.items():
    if k.startswith('k_l4a') or 'k_wave<3>' in k or 'k_final' in k:
        g=lambda n: v.get(n,{}).get('per_launch',0)
        cyc=g('GRBM_GUI_ACTIVE')/8
        print(k, 'cycles/XCD %.2fM'%(cyc/1e6), 'VALU %.0fM'%(g('SQ_INSTS_VALU')/1e6), 'SALU %.0fM'%(g('SQ_INSTS_SALU')/1e6), 'LDS %.0fM'%(g('SQ_INSTS_LDS')/1e6), 'VMEM %.1fM'%(g('SQ_INSTS_VMEM')/1e6),
              'valu_util %.2f'%(g('SQ_INSTS_VALU')*4/(1024*cyc) if cyc else 0), 'salu_util %.2f'%(g('SQ_INSTS_SALU')/(256*cyc) if cyc else 0), 'wave_cycles %.0fM'%(g('SQ_WAVE_CYCLES')/1e6), 'wait_any %.0fM'%(g('SQ_WAIT_ANY')/1e6), 'wait_inst %.0fM'%(g('SQ_WAIT_INST_ANY')/1e6), 'active_valu %.0fM'%(g('SQ_ACTIVE_INST_VALU')/1e6), 'bank_conf %.0fM'%(g('SQ_LDS_BANK_CONFLICT')/1e6), 'lds_active %.0fM'%(g('SQ_ACTIVE_INST_LDS')/1e6))
PY
