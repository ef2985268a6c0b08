#!/bin/bash
# developer ablation: rebuild postproc.hip with -D flags on the GPU box and report per-kernel times of the nuclei pipeline at 8192^2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
IFS=';'
for FL in ${CERB_VARIANTS:-""}; do
  unset IFS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -c cerberus_amd/csrc/postproc.hip -o cerberus_amd/csrc/postproc.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cerberus_amd/libcerberus_hip.so cerberus_amd/csrc/*.o || exit 1
  echo "=== flags: [$FL]"
  python scripts/dev_pp_nuclei_only.py ${PP_SIDE:-8192} 2>&1 | tail -1
  rm -rf gpurun_out/pp_prof
  rocprofv3 --kernel-trace --stats -d gpurun_out/pp_prof -o pp -- python scripts/dev_pp_nuclei_only.py ${PP_SIDE:-8192} > /dev/null 2>&1
  python scripts/rocprof_summary.py stats "$(find gpurun_out/pp_prof -name '*.db' | head -1)" gpurun_out/pp_kernel_stats.txt > /dev/null
  grep -E "${PP_GREP:-ccl_|erode|threshold|apply_min|ws_seed|ws_bbox}" gpurun_out/pp_kernel_stats.txt | head -14 | cut -c1-60,95-140
  rm -rf gpurun_out/pp_prof
  IFS=';'
done
