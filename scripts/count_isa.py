#!/usr/bin/env python3
"""DEV-ONLY: instruction mix of the main kernels from the gfx950 assembly.
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -ffp-contract=on --cuda-device-only -S -o k.s csrc/hydro_kernels.hip
python scripts/count_isa.py k.s"""
import re, sys
txt = open(sys.argv[1]).read()
names = sys.argv[2:] or ['wrench_tiled_kernelILi256ELb1ELb0ELb1E', 'wrench_tiled_kernelILi256ELb0ELb0ELb1E',
                         'step_fused_tiled_kernelILb1ELb0ELb1E', 'wrench_aos_kernelILb0ELb1E']
for name in names:
    m = re.search(r'^(_ZN\S*' + name + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S | re.M)
    if not m:
        print(name, 'not found'); continue
    ins = [l.strip().split()[0] for l in m.group(2).split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    valu = [i for i in ins if i.startswith('v_')]
    print(f"{name}: total {len(ins)} valu {len(valu)} f64 {sum('f64' in i for i in valu)} "
          f"sqrt/rcp {sum(('sqrt' in i or 'rcp' in i) for i in valu)} loads {sum(i.startswith('global_load') for i in ins)} "
          f"stores {sum(i.startswith('global_store') for i in ins)} salu {sum(i.startswith('s_') for i in ins)} lds {sum(i.startswith('ds_') for i in ins)}")
