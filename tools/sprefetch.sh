#!/bin/bash
# scalar-path prefetch into the L2: rates and whether a vector stream behind it runs at L2-hit speed (tools/sprefetch.hip)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=build/tools/sprefetch
echo "== scalar touches only (s_load_dword per 128-byte line, 15 in flight per wave), 256 workgroups, 1 MiB per wave"
for w in 1 2 4 8 16; do $B 0 $w 256 1048576 | tail -1; done
echo "== vector reads only (buffer_load_dwordx4 nt), 256 workgroups"
for w in 2 8 16; do $B 1 $w 256 1048576 | tail -1; done
echo "== 8 waves read chunk k (vector, nt) while 8 waves touch chunk k + ahead (scalar); 128 KiB per workgroup and step, 64 steps; touch off / on"
$B 2 0 256 0 1 0 | tail -1
for a in 1 2 4; do $B 2 0 256 0 $a 1 | tail -1; done
echo "== the same on 512 workgroups (two per CU)"
$B 2 0 512 0 1 0 | tail -1
$B 2 0 512 0 2 1 | tail -1
