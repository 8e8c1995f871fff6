#!/bin/bash
# VERDICT r4 #2, timing-only probe: can the matrix work of the aggregate -> contract fusion hide under the HBM-bound gather
# on the SAME CUs?  Builds libraries whose gather issues N dummy v_mfma_f32_32x32x16_f16 per four gathered rows (the
# contraction Z_r W_r of a dim-256, 16-level layer needs 6.25) from a PATCHED COPY of seg_gather.hip -- the shipped source
# (and the sha its PMC records are stamped with) is not touched.  tools/ablate/fp_<N>/libstargcn_hip.so
set -e
cd "$(dirname "$0")/../star-gcn_amd/csrc"
make -j8 > /dev/null
OBJS="stream_read.o seg_ops.o gemm_f32.o gemm_bf16x6.o gemm_x6v2.o gemm_f16x3.o gemm_x3w.o multilink.o edge_mask.o embed.o plan_build.o graph_host.o"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I."
python3 - <<'PY'
s = open('seg_gather.hip').read()
old = '''      ld_row<VEC>(x[u], src + off + c);      // unconditional (c is clamped by the caller): see the note above the function
    }
'''
new = old + '''#ifdef SG_GATHER_DUMMY_MFMA
    if (VEC == 4) {      // timing only: the contraction's matrix instructions, on dummy operands, two accumulation chains
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      typedef float fv4 __attribute__((ext_vector_type(4)));
      const fv4 p0 = {x[0][0], x[0][1], x[0][2], x[0][3]}, p1 = {x[1][0], x[1][1], x[1][2], x[1][3]};
      const h8 fa = __builtin_bit_cast(h8, p0), fb = __builtin_bit_cast(h8, p1);
#pragma unroll
      for (int q = 0; q < SG_GATHER_DUMMY_MFMA; ++q) {
        if (q & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(g_dm1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(g_dm0) : "v"(fa), "v"(fb));
      }
    }
#endif
'''
assert old in s
s = s.replace(old, new, 1)
old2 = '''#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  int e = ea + grp;'''
new2 = '''#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#ifdef SG_GATHER_DUMMY_MFMA
  typedef float f16v __attribute__((ext_vector_type(16)));
  f16v g_dm0, g_dm1;
  asm volatile("" : "=v"(g_dm0), "=v"(g_dm1));
#endif
  int e = ea + grp;'''
assert old2 in s
s = s.replace(old2, new2, 1)
old3 = '''  if (!UNI) {
    if (DOT) {
#pragma unroll
      for (int off = (DLPR > 0 ? DLPR : kWave); off < kWave; off <<= 1) {'''
new3 = '''#ifdef SG_GATHER_DUMMY_MFMA
  asm volatile("s_nop 15\\n\\ts_nop 3" : : "v"(g_dm0), "v"(g_dm1));
#endif
''' + old3
assert old3 in s
s = s.replace(old3, new3, 1)
open('/tmp/seg_gather_probe.hip', 'w').write(s)
PY
for n in 6 12; do
  mkdir -p ../../tools/ablate/fp_$n
  /opt/rocm/bin/hipcc $FLAGS -DSG_GATHER_DUMMY_MFMA=$n -c /tmp/seg_gather_probe.hip -o /tmp/seg_gather_fp$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ablate/fp_$n/libstargcn_hip.so $OBJS /tmp/seg_gather_fp$n.o
done
ls -la ../../tools/ablate/fp_*/libstargcn_hip.so
