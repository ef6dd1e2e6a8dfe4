#!/bin/sh
# Prepares the reference's OWN CUDA sources for g++ (oracle/Makefile target `ref_cu`).  TEST INFRASTRUCTURE ONLY.
# Reads /root/reference where it lies, writes ONLY into oracle/_ref/gen/ (git-ignored, deleted after the build).
#
# The edits are mechanical and listed here in full -- everything else is compiled as written:
#  (1) every `kernel<<<grid, block[, shmem, stream]>>>(args);` becomes `SHIM_LAUNCH(kernel, (grid, block...), (args));`
#      (cuda_shim/cuda_runtime_api.h), because `<<< >>>` is not C++;
#  (2) tsdf_volume.cu only: `__CUDA_ARCH__` -> `SHIM_CUDA_ARCH` (= 300), so FullScan6 takes its __ballot / __all path as on
#      any GPU the reference targets, while device.hpp keeps its plain-C++ LdCs/StCs (its PTX path sits behind the real macro);
#  (3) lock-step patch for the two pieces of WARP-SYNCHRONOUS code (32 lanes executing one statement together, readers
#      before writers).  A host fiber executes a statement alone, so such a statement is split into
#      `sync; read phase; sync; write phase; sync` with cuda_shim::sync_warp() -- the same values in the same order:
#        - Block::reduce's warp tail `{ buffer[tid] = val = op(val, buffer[tid + N]); }`      (temp_utils.hpp:509-517,536-541)
#        - scan_warp's `if (lane >= N) ptr[idx] = ptr[idx - N] + ptr[idx];`                    (tsdf_volume.cu:493-497)
#        - FullScan6's aliased cta_buffer / storage_X hand-over                                 (tsdf_volume.cu:646-668)
#      and Warp::laneId()'s PTX `%laneid` read becomes cuda_shim::lane_id()                     (temp_utils.hpp:451-456);
#  (4) the three device-struct constructors of precomp.cpp:24-25,42,55 and ComputeIcpHelper's constructor + setLevelIntr
#      (projective_icp.cpp:11-23) are cut out by pattern into device_ctors.cpp (their files need OpenCV as a whole).
set -e
REF=${1:-/root/reference}
GEN=${2:-_ref/gen}
SRC=$REF/kfusion/src
mkdir -p "$GEN"

LAUNCH='s/^([[:space:]]*)([A-Za-z_][A-Za-z_0-9]*)[[:space:]]*<<<(.*)>>>[[:space:]]*\((.*)\);/\1SHIM_LAUNCH(\2, (\3), (\4));/'
SYNC='::cuda_shim::sync_warp();'

sed -E \
    -e "$LAUNCH" \
    -e 's/__CUDA_ARCH__/SHIM_CUDA_ARCH/g' \
    -e "s/^([[:space:]]*)if \(lane >= +([0-9]+)\) ptr\[idx\] = ptr\[idx - +[0-9]+\] \+ ptr\[idx\];/\1{ $SYNC T t_ = ptr[idx]; if (lane >= \2) t_ = ptr[idx - \2] + ptr[idx]; $SYNC ptr[idx] = t_; $SYNC }/" \
    -e "s/^([[:space:]]*int offset = scan_warp<exclusive>\(cta_buffer, lane\);)/\1 $SYNC/" \
    -e "s/^([[:space:]]*)(int old_global_count = cta_buffer\[0\];)/\1$SYNC \2 $SYNC/" \
    -e "s/^([[:space:]]*)(Point \*pos = output\.data \+ old_global_count \+ lane;)/\1$SYNC \2/" \
    "$SRC/cuda/tsdf_volume.cu" > "$GEN/tsdf_volume.cu.cpp"
sed -E -e "$LAUNCH" "$SRC/cuda/imgproc.cu" > "$GEN/imgproc.cu.cpp"
sed -E -e "$LAUNCH" "$SRC/cuda/proj_icp.cu" > "$GEN/proj_icp.cu.cpp"

sed -E \
    -e "s/\{ buffer\[tid\] = val = op\(val, buffer\[tid \+ +([0-9]+)\]\); \}/{ $SYNC T t_ = op(val, buffer[tid + \1]); $SYNC buffer[tid] = val = t_; $SYNC }/" \
    -e 's/asm\("mov\.u32 %0, %laneid;" : "=r"\(ret\) \);/ret = ::cuda_shim::lane_id();/' \
    "$SRC/cuda/temp_utils.hpp" > "$GEN/temp_utils.hpp"
# device.hpp / texture_binder.hpp unchanged; copied only so that device.hpp's `#include "temp_utils.hpp"` finds the patched one
cp "$SRC/cuda/device.hpp" "$SRC/cuda/texture_binder.hpp" "$GEN/"

{
    echo '#include "cuda_runtime_api.h"'
    echo '#include <cmath>'
    echo '#include "internal.hpp"'
    echo 'using std::cos;'
    sed -n '/^kfusion::device::TsdfVolume::TsdfVolume(elem_type/,/{}/p' "$SRC/precomp.cpp"
    grep -E '^kfusion::device::(Projector::Projector|Reprojector::Reprojector)\(float fx' "$SRC/precomp.cpp"
    sed -n '/^kfusion::device::ComputeIcpHelper::ComputeIcpHelper(/,/^}/p' "$SRC/projective_icp.cpp"
    sed -n '/^void kfusion::device::ComputeIcpHelper::setLevelIntr(/,/^}/p' "$SRC/projective_icp.cpp"
} > "$GEN/device_ctors.cpp"

# sanity: every edit must have hit (the reference is pinned; a silent miss would leave unsynchronised code)
chk() { n=$(grep -c "$2" "$1" || true); [ "$n" -ge "$3" ] || { echo "gen_ref_cu: expected >= $3 of '$2' in $1, found $n" >&2; exit 1; }; }
chk "$GEN/tsdf_volume.cu.cpp" 'SHIM_LAUNCH' 8
chk "$GEN/tsdf_volume.cu.cpp" 'sync_warp' 8
chk "$GEN/imgproc.cu.cpp" 'SHIM_LAUNCH' 14
chk "$GEN/proj_icp.cu.cpp" 'SHIM_LAUNCH' 4
chk "$GEN/temp_utils.hpp" 'sync_warp' 12
chk "$GEN/temp_utils.hpp" 'cuda_shim::lane_id' 1
chk "$GEN/device_ctors.cpp" '^kfusion::device::' 4
if grep -n '<<<' "$GEN"/*.cu.cpp; then echo "gen_ref_cu: unconverted kernel launch" >&2; exit 1; fi
