#!/bin/bash
# How much does k_trace's time move per extra L1 look-up and per extra vector instruction of a node step?  (profiles/r04_node32_estimate.txt:
# a 32-byte node trades -2 look-ups for ~+55 instructions per step; this measures both slopes on the benchmark launch instead of assuming them.)
#   build (here, no GPU needed):  bash tools/sensitivity_probe.sh build     -> tools/libgvt_hip_probe_{C0,L1,L2,L4,V32,V64}.so from a patched COPY of csrc/
#   run   (GPU box):              bash tools/sensitivity_probe.sh run       -> gpurun_out/sensitivity.txt
# The shipped sources are not touched: the patch below is applied to a scratch copy.
#   L<n>: n extra `global_load_lds` dword loads of the node's own line per node step (destination LDS: no registers, the line is the one the
#         node fetch brings anyway -> no extra HBM / L2 traffic, only n x 64 more look-ups in the L1's tag pipe per wave step; a node fetch is 4 x 64)
#   C0:   the look-up probe's M0 write and compiler barrier without a load (control)
#   V<n>: n extra `v_mov_b32 cur, cur` per node step (no registers, no memory: pure vector issue)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ "$1" = build ]; then
  for v in ${PROBES:-C0 L1 L2 L4 V32 V64}; do
    W=$(mktemp -d /tmp/gvt_probe_XXXX)
    cp -r gravit_amd/csrc/*.hip gravit_amd/csrc/*.h gravit_amd/csrc/*.inc "$W"/
    python3 - "$W/trace_lane.inc" <<'EOF'
import sys
p = sys.argv[1]
s = open(p).read()
anchor = "          node4_test((MULTI ? nodes4_l : T.nodes4) + (size_t)GVT_NODE4_F4 * cur, S,"
assert s.count(anchor) == 1
probe = r'''#if KT_EXTRA_LOOKUPS
          { __shared__ unsigned s_sink_[64]; // (the builtin of the same name does not keep its LDS operand alive: explicit M0 + instruction; k_trace uses M0 nowhere else)
            if (n == 0xffffffffu) ((volatile unsigned *)s_sink_)[0] = 0;
            const unsigned base_ = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)s_sink_;
            const unsigned *p_ = (const unsigned *)((MULTI ? nodes4_l : T.nodes4) + (size_t)GVT_NODE4_F4 * cur);
#pragma unroll
            for (int e_ = 0; e_ < KT_EXTRA_LOOKUPS; e_++) { const unsigned *q_ = p_ + 4 * (e_ & 3); asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"(base_), "v"(q_) : "memory"); } }
#endif
#if KT_PROBE_CONTROL // the look-up probe without its loads: the M0 write and the compiler barrier alone
          { __shared__ unsigned s_sink_[64];
            if (n == 0xffffffffu) ((volatile unsigned *)s_sink_)[0] = 0;
            const unsigned base_ = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)s_sink_;
            asm volatile("s_mov_b32 m0, %0" :: "s"(base_) : "memory"); }
#endif
#if KT_EXTRA_VALU
#pragma unroll
          for (int e_ = 0; e_ < KT_EXTRA_VALU; e_++) asm volatile("v_mov_b32 %0, %0" : "+v"(cur));
#endif
'''
open(p, "w").write(s.replace(anchor, probe + anchor))
EOF
    case $v in C*) D="-DKT_PROBE_CONTROL=1";; L*) D="-DKT_EXTRA_LOOKUPS=${v#L}";; V*) D="-DKT_EXTRA_VALU=${v#V}";; esac
    objs=""
    for s in api lbvh trace sched domain; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I"$ROOT/include" -I"$ROOT/gravit_amd/csrc" $D -c "$W/$s.hip" -o "$W/$s.o" &
      objs="$objs $W/$s.o"
    done
    wait
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "tools/libgvt_hip_probe_$v.so" $objs
    rm -rf "$W"
    echo "built tools/libgvt_hip_probe_$v.so"
  done
  exit 0
fi
mkdir -p gpurun_out
OUT=gpurun_out/sensitivity.txt
: > $OUT
for v in shipped C0 L1 L2 L4 V32 V64 shipped; do
  lib=gravit_amd/libgvt_hip.so
  [ $v != shipped ] && lib=tools/libgvt_hip_probe_$v.so
  echo "== $v" >> $OUT
  GVT_HIP_LIB=$ROOT/$lib python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-abi-path --no-sustained > gpurun_out/sens_$v.log 2>&1
  python3 - gpurun_out/sens_$v.log >> $OUT <<'EOF'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(json.dumps({"ms_per_step": j["ms_per_step"], "roofline": j["roofline"], "kernels": j.get("kernel_ms_per_step") or j.get("launch_ms")}))
EOF
done
cat $OUT
