#!/bin/bash
# builds build/exp/libs/libwrp_a_entry.so: the stamps instantiation of fused_chain_1024x512 with three more stamps per workgroup
# (entry into the kernel, end of the team meeting, exit) for tools/fused_launch_edges.py.  Run from build/exp (see var.sh).
cd /root/repo/build/exp && ./var.sh a_entry '
rep("""    const int n = DP_N, gates = RP_M / 2;

    const FusedSeat seat = fused_join(ctl, s_ctl);""","""    const int n = DP_N, gates = RP_M / 2;
    const unsigned long long t_entry = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
    const FusedSeat seat = fused_join(ctl, s_ctl);
    const unsigned long long t_joined = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;""", file="wrp_fused.h")
rep("""        if (tid == 0) s_stamps[FUSED_STAMPS - 1] = ((unsigned long long)kind << 32) | ((unsigned long long)xcc << 16) | (unsigned)rank;""","""        if (tid == 0) { s_stamps[FUSED_STAMPS - 1] = ((unsigned long long)kind << 32) | ((unsigned long long)xcc << 16) | (unsigned)rank;
            s_stamps[3 * FUSED_STAMPS + 8] = t_entry; s_stamps[4 * FUSED_STAMPS + 8] = t_joined; }""", file="wrp_fused.h")
rep("""                s_stamps[2 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;""","""                s_stamps[2 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
                s_stamps[5 * FUSED_STAMPS + 8] = __builtin_amdgcn_s_memrealtime();""", file="wrp_fused.h")
'
