#!/bin/bash
# compile tools/isa_scatter.hip for gfx950 with -save-temps and print the resource usage of every kernel in it
set -e
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I/root/repo/include \
  -I/root/repo/neuralvolumetricreconstructionformedicalimages_amd/csrc -c /root/repo/tools/${1:-isa_scatter}.hip -o s.o -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage 2> s.remarks || { grep -v "remark:" s.remarks | head -40; exit 1; }
grep -v "remark:" s.remarks | head -20
grep "Function Name\| VGPRs:\|SGPRs:\|Spill\|Occupancy" s.remarks | sed 's/.*remark: [^ ]* *//' | sed 's/\[-Rpass.*//' | paste - - - - - - | sed 's/Function Name: //' | awk '{print substr($1,1,90), $2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15}'
