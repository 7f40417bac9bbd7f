#!/bin/bash
# usage: mk.sh NAME [extra hipcc flags...]  -> build/exp/libs/libwrp_NAME.so from the CURRENT csrc (or $SRC), prints registers / spills of the fused kernels
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -I/root/repo/include -I/root/repo/weather-radar-processing_amd -Rpass-analysis=kernel-resource-usage -shared -o /root/repo/build/exp/libs/libwrp_$name.so ${SRC:-/root/repo/weather-radar-processing_amd/csrc}/wrp_engine.hip "$@" 2> /root/repo/build/exp/libs/$name.remarks
python3 /root/repo/build/exp/regs.py /root/repo/build/exp/libs/$name.remarks ${PAT:-fused_chain}
