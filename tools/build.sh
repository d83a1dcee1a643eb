#!/bin/bash
# Builds the kernel A/B harness and ablation variants of the library into tools/_build (not product code).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
F="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden"
hipcc -O2 -std=c++17 -o tools/_build/kbench tools/kbench.cpp -ldl
for v in "$@"; do
  case $v in
    nomath)   hipcc $F -DGF_ABL_NOMATH   -o tools/_build/lib_nomath.so   gyroflow_amd/csrc/gfremap.hip ;;
    nogather) hipcc $F -DGF_ABL_NOGATHER -o tools/_build/lib_nogather.so gyroflow_amd/csrc/gfremap.hip ;;
    noload) hipcc $F -DGF_ABL_NOLOAD -o tools/_build/lib_noload.so gyroflow_amd/csrc/gfremap.hip ;;
    nostore) hipcc $F -DGF_ABL_NOSTORE -o tools/_build/lib_nostore.so gyroflow_amd/csrc/gfremap.hip ;;
    nobarrier) hipcc $F -DGF_ABL_NOBARRIER -o tools/_build/lib_nobarrier.so gyroflow_amd/csrc/gfremap.hip ;;
    noload_nostore) hipcc $F -DGF_ABL_NOLOAD -DGF_ABL_NOSTORE -o tools/_build/lib_noload_nostore.so gyroflow_amd/csrc/gfremap.hip ;;
    k1_*) # k1_<pipeline>_<waves>, e.g. k1_0_8
        IFS=_ read -r _ pl wv <<< "$v"
        hipcc $F -DGF_K1_PIPELINE=$pl -DGF_K1_WAVES=$wv -o tools/_build/lib_$v.so gyroflow_amd/csrc/gfremap.hip ;;
    tile_*) # tile_<W>_<H>
        IFS=_ read -r _ tw th <<< "$v"
        hipcc $F -DGF_TILE_W=$tw -DGF_TILE_H=$th -o tools/_build/lib_$v.so gyroflow_amd/csrc/gfremap.hip ;;
    pairs_*) hipcc $F -DGF_COORD_PAIRS=1 -DGF_K1_WAVES=${v#pairs_} -o tools/_build/lib_$v.so gyroflow_amd/csrc/gfremap.hip ;;
    trace)    hipcc $F -DGF_TRACE_WG -o tools/_build/lib_trace.so gyroflow_amd/csrc/gfremap.hip gyroflow_amd/csrc/quat_host.cpp ;;   # kbench --lib prints the workgroup timeline
    pf_*)     hipcc $F -DGF_RP_PREFETCH=${v#pf_} -o tools/_build/lib_$v.so gyroflow_amd/csrc/gfremap.hip gyroflow_amd/csrc/quat_host.cpp ;;     # K1 touching its rows <n> source rows ahead
    lb_*)     hipcc $F -DGF_LB -DGF_LB_WAVES=${v#lb_} -o tools/_build/lib_$v.so gyroflow_amd/csrc/gfremap.hip gyroflow_amd/csrc/quat_host.cpp ;;
    lb)       hipcc $F -DGF_LB -o tools/_build/lib_lb.so gyroflow_amd/csrc/gfremap.hip gyroflow_amd/csrc/quat_host.cpp ;;      # mode rl (GF_K1=rl) runs the half-tile LDS-box kernel
    both)     hipcc $F -DGF_ABL_NOMATH -DGF_ABL_NOGATHER -o tools/_build/lib_both.so gyroflow_amd/csrc/gfremap.hip ;;
  esac
done
