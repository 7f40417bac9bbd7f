#!/bin/bash
# usage: var.sh NAME 'python expr on s (file text of wrp_fused_b.h / wrp_fused.h)' [hipcc flags]  -- builds libs/libwrp_NAME.so from a patched copy of csrc
name=$1; edit=$2; shift 2
r=/root/repo/build/exp/src/$name; rm -rf $r; d=$r/pkg/csrc; mkdir -p $d $r/include; cp /root/repo/include/wrp.h $r/include/; cp /root/repo/weather-radar-processing_amd/csrc/* $d/
python3 - "$d" "$edit" <<'PY'
import sys,re
d,edit=sys.argv[1],sys.argv[2]
for f in ('wrp_fused_b.h','wrp_fused.h','wrp_kernels.h','fft_radix.h','wrp_shape_b.h'):
    p=d+'/'+f; s=open(p).read(); s0=s
    def rep(a,b,count=1,file=None):
        global s
        if file and file!=f: return
        if a in s: s=s.replace(a,b,count)
    exec(edit)
    if s!=s0: open(p,'w').write(s); print("patched",f)
PY
SRC=$d PAT=${PAT:-fused_chain_2048x128<7, false, false} ./mk.sh $name "$@"
