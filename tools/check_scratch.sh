#!/bin/bash
# Lists every kernel of the library that uses scratch (private) memory: tools/check_scratch.sh [file.hip ...]
# (device-only -S of each csrc/*.hip, in parallel; CPU only, a few minutes).  Expected: nothing but the 20-36 byte tables of the
# legacy fp32 / round-1 GEMM kernels in air_gemm.hip.  A hot kernel that shows up here has an array the compiler could not keep
# in registers (HIP vector structs carried across barriers) or ran out of its register budget
# (wgrad_grouped_bf16_kernel sits at exactly 168).
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/tf-attend-infer-repeat_amd/csrc
files=("$@"); [ ${#files[@]} -eq 0 ] && files=($src/*.hip)
tmp=$(mktemp -d)
for f in "${files[@]}"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$root/include -I$src --cuda-device-only -S "$f" -o $tmp/$(basename $f).s 2>/dev/null ) &
done
wait
for s in $tmp/*.s; do
  awk -v f=$(basename $s .s) '/^_Z.*:/{name=$1} /; NumVgprs:/{v=$3} /; ScratchSize: [1-9]/{printf "%s  %s scratch %s bytes, %s VGPRs\n", f, name, $3, v}' $s
done | c++filt | sed 's/(anonymous namespace):://'
rm -rf $tmp
