/* simd_baseline.c -- the SIMD CPU baseline of bench.py's `cpu_baseline` leg.  MEASUREMENT INFRASTRUCTURE, like the rest of oracle/: nothing
 * under gravit_amd/ links, loads or calls it (tests/test_host_cpu.py checks); it is NOT the oracle either -- the oracle (gvt_oracle.c) is the
 * scalar restatement results are checked against, this file is the thing that is TIMED beside the GPU number, and its own hits are validated
 * against the oracle's before its time is reported (bench.py, tests/test_gpu_parity.py).
 *
 * Why it exists: the reference's CPU path is Embree 2.x -- a SAH BVH4/8 walked one ray (or one packet) against four boxes at a time in SSE / AVX
 * (rtcIntersect4/8/16, rtcOccluded: src/gvt/render/adapter/embree/EmbreeMeshAdapter.cpp:474, :375; the packet loop :436-660) -- and Embree is not in
 * the tree (un-vendored submodule).  The scalar median-split BVH2 of the oracle stands one to two orders below what the host does with
 * such a traversal.  This is the closest stand-in buildable here: the GPU-built compressed 4-wide tree, downloaded once
 * (gvt_hip_mesh_download_wide), walked by ONE RAY AGAINST FOUR CHILD BOXES per step with SSE4.1 + FMA (the 8-bit child planes widened with
 * pmovzxbd, six fused multiply-adds for the slabs), nearest child first, a stack per ray, leaves of <= 2 triangles tested with the restated
 * Moeller-Trumbore test of the kernels (strict IEEE, no contraction: the same (t, primID, u, v) bits as the oracle and the kernels -- a box test
 * cannot change a result, boxes are conservative), rays in chunks of 4096 over pthreads like the reference's TBB grain (:648).
 *
 * Node and slot layouts: include/gvt_hip.h, gvt_hip_mesh_download_wide.  gcc -O3 -msse4.1 -mfma -ffp-contract=off (oracle/Makefile). */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define SB_FLT_MAX 3.402823466e+38f
#define SB_STACK 256
#define SB_CHUNK 4096

typedef struct {
  const uint32_t *nodes4; /* 16 words per node */
  const float *slots;     /* 16 floats per triangle slot */
  int has_nodes;
  const float *org, *dir; /* n x 3, object space */
  size_t n;
  float tnear;
  int any;                /* 1: occluded(), 0: intersect() */
  float *t, *u, *v;       /* closest hit */
  int32_t *prim;
  uint8_t *occ;           /* any hit */
  atomic_size_t next;
  atomic_ullong node_steps, leaf_steps, overflow;
} Job;

typedef struct { float x, y, z; } V3;
static inline V3 mk3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
static inline V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float dot3(V3 a, V3 b) { const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }
static inline V3 cross3(V3 x, V3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }

/* the kernels' triangle test (csrc/gvt_device.h tri_test_raw), operation for operation */
static inline int tri_test_raw(V3 O, V3 D, V3 v0, V3 e1, V3 e2, V3 Ng, float tnear, float *T, float *U, float *V, float *absDen) {
  const V3 C = sub3(v0, O);
  const V3 R = cross3(D, C);
  const float den = dot3(Ng, D);
  *absDen = fabsf(den);
  const float sgn = (den < 0.f) ? -1.f : 1.f;
  *U = dot3(R, e2) * sgn;
  *V = dot3(R, e1) * sgn;
  if (!(den != 0.f && *U >= 0.f && *V >= 0.f && *U + *V <= *absDen)) return 0;
  *T = dot3(Ng, C) * sgn;
  return *absDen * tnear < *T;
}

/* the cull distance: the best t with the relative slack of the kernels and the oracle (csrc/gvt_device.h cull_bound, gvt_oracle.c box_test) */
static inline float sb_cull(float bt) { return fmaf(fabsf(bt), 0x1p-10f, bt); }

static inline __m128 q4(uint32_t q) { return _mm_cvtepi32_ps(_mm_cvtepu8_epi32(_mm_cvtsi32_si128((int)q))); } /* four 8-bit planes -> four floats */

static void trace_one(const Job *J, size_t j, unsigned long long *n_node, unsigned long long *n_leaf) {
  const V3 O = mk3(J->org[3 * j], J->org[3 * j + 1], J->org[3 * j + 2]), D = mk3(J->dir[3 * j], J->dir[3 * j + 1], J->dir[3 * j + 2]);
  const float dx = fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x, dy = fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y,
              dz = fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z;
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
  const float ox = O.x * ix, oy = O.y * iy, oz = O.z * iz;
  /* slabs widened like the kernels' RaySlab (trace_lane.inc make_slab): near offsets moved out, far offsets in, by 2^-21 |O / d| */
  const float ex = fabsf(ox) * 4.76837158e-7f, ey = fabsf(oy) * 4.76837158e-7f, ez = fabsf(oz) * 4.76837158e-7f;
  const float oxn = ox + ex, oxf = ox - ex, oyn = oy + ey, oyf = oy - ey, ozn = oz + ez, ozf = oz - ez;
  float bt = SB_FLT_MAX, bu = 0.f, bv = 0.f, bden = 1.f;
  int bp = -1;
  int32_t st_ref[SB_STACK];
  float st_tn[SB_STACK];
  int sp = 0;
  int32_t cur = J->has_nodes ? 0 : INT32_MIN;
  int done = !J->has_nodes;
  /* the next stack entry that can still hold something nearer than the best hit (any hit: the next entry) */
#define SB_POP() { cur = INT32_MIN; while (sp) { sp--; if (J->any || st_tn[sp] <= sb_cull(bt)) { cur = st_ref[sp]; break; } } if (cur == INT32_MIN) break; }
  while (!done) {
    if (cur >= 0) {
      (*n_node)++;
      const uint32_t *nd = J->nodes4 + 16 * (size_t)cur;
      float w0[4], w3f[4];
      memcpy(w0, nd, 16); memcpy(w3f, nd + 12, 16);
      const float sx = w0[3] * ix, sy = w3f[2] * iy, sz = w3f[3] * iz;
      const float bxn = fmaf(w0[0], ix, -oxn), bxf = fmaf(w0[0], ix, -oxf), byn = fmaf(w0[1], iy, -oyn), byf = fmaf(w0[1], iy, -oyf),
                  bzn = fmaf(w0[2], iz, -ozn), bzf = fmaf(w0[2], iz, -ozf);
      const uint32_t qnx = ix >= 0.f ? nd[4] : nd[5], qfx = ix >= 0.f ? nd[5] : nd[4], qny = iy >= 0.f ? nd[6] : nd[7], qfy = iy >= 0.f ? nd[7] : nd[6],
                     qnz = iz >= 0.f ? nd[8] : nd[9], qfz = iz >= 0.f ? nd[9] : nd[8];
      const __m128 nx = _mm_fmadd_ps(q4(qnx), _mm_set1_ps(sx), _mm_set1_ps(bxn)), ny = _mm_fmadd_ps(q4(qny), _mm_set1_ps(sy), _mm_set1_ps(byn)),
                   nz = _mm_fmadd_ps(q4(qnz), _mm_set1_ps(sz), _mm_set1_ps(bzn));
      const __m128 fx = _mm_fmadd_ps(q4(qfx), _mm_set1_ps(sx), _mm_set1_ps(bxf)), fy = _mm_fmadd_ps(q4(qfy), _mm_set1_ps(sy), _mm_set1_ps(byf)),
                   fz = _mm_fmadd_ps(q4(qfz), _mm_set1_ps(sz), _mm_set1_ps(bzf));
      const __m128 tnr = _mm_max_ps(_mm_max_ps(nx, ny), _mm_max_ps(nz, _mm_setzero_ps()));
      const __m128 tfr = _mm_mul_ps(_mm_min_ps(_mm_min_ps(fx, fy), fz), _mm_set1_ps(1.0000004f));
      const __m128 lim = _mm_set1_ps(J->any ? SB_FLT_MAX : sb_cull(bt));
      const int mask = _mm_movemask_ps(_mm_cmple_ps(tnr, _mm_min_ps(tfr, lim)));
      if (!mask) { SB_POP(); continue; }
      float tn[4];
      _mm_storeu_ps(tn, tnr);
      const int32_t rr[4] = { (int32_t)nd[10], (int32_t)nd[11], (int32_t)nd[12], (int32_t)nd[13] };
      /* the entered children nearest first (insertion into a list of at most four) */
      int idx[4], k = 0;
      for (int c = 0; c < 4; c++) {
        if (!(mask >> c & 1)) continue;
        int p = k++;
        while (p > 0 && tn[idx[p - 1]] > tn[c]) { idx[p] = idx[p - 1]; p--; }
        idx[p] = c;
      }
      for (int p = k - 1; p >= 1; p--) {
        if (sp == SB_STACK) { atomic_fetch_add(&((Job *)J)->overflow, 1ull); done = 1; break; } /* reported: the call fails (the GPU kernels' own stack has 24 + spill levels for the same trees) */
        st_ref[sp] = rr[idx[p]]; st_tn[sp] = tn[idx[p]]; sp++;
      }
      cur = rr[idx[0]];
    } else {
      (*n_leaf)++;
      const uint32_t code = (uint32_t)~cur;
      const uint32_t first = code >> 3, ntri = code & 7u;
      for (uint32_t t = 0; t < ntri; t++) {
        const float *s = J->slots + 16 * (size_t)(first + t);
        const V3 e1 = mk3(s[4], s[5], s[6]), e2 = mk3(s[8], s[9], s[10]);
        float TT, U, V, aden;
        if (tri_test_raw(O, D, mk3(s[0], s[1], s[2]), e1, e2, cross3(e1, e2), J->tnear, &TT, &U, &V, &aden)) {
          const float tt = TT / aden;
          if (tt <= SB_FLT_MAX) {
            int32_t prim;
            memcpy(&prim, s + 3, 4);
            if (J->any) { bp = 0; done = 1; break; }
            if (bp < 0 || tt < bt || (tt == bt && prim < bp)) { bt = tt; bp = prim; bu = U; bv = V; bden = aden; }
          }
        }
      }
      if (done) break;
      SB_POP();
    }
  }
#undef SB_POP
  if (J->any) J->occ[j] = bp >= 0 ? 1 : 0;
  else { J->t[j] = bt; J->prim[j] = bp; J->u[j] = bp >= 0 ? bu / bden : 0.f; J->v[j] = bp >= 0 ? bv / bden : 0.f; }
}

static void *worker(void *arg) {
  Job *J = (Job *)arg;
  unsigned long long nn = 0, nl = 0;
  for (;;) {
    const size_t a = atomic_fetch_add(&J->next, (size_t)SB_CHUNK);
    if (a >= J->n) break;
    const size_t b = a + SB_CHUNK < J->n ? a + SB_CHUNK : J->n;
    for (size_t j = a; j < b; j++) trace_one(J, j, &nn, &nl);
  }
  atomic_fetch_add(&J->node_steps, nn);
  atomic_fetch_add(&J->leaf_steps, nl);
  return NULL;
}

static int run(Job *J, int nthreads, unsigned long long *steps /* [2] or NULL */) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  atomic_init(&J->next, 0); atomic_init(&J->node_steps, 0); atomic_init(&J->leaf_steps, 0); atomic_init(&J->overflow, 0);
  pthread_t th[256];
  int started = 0;
  for (int k = 0; k < nthreads - 1; k++) { if (pthread_create(&th[started], NULL, worker, J) == 0) started++; }
  worker(J);
  for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
  if (steps) { steps[0] = atomic_load(&J->node_steps); steps[1] = atomic_load(&J->leaf_steps); }
  return atomic_load(&J->overflow) ? -1 : 0;
}

/* closest hits (rtcIntersect): t = FLT_MAX, prim = -1 on a miss.  org / dir: n x 3 floats in the mesh's object space. */
int simd_intersect(const uint32_t *nodes4, size_t n_nodes4, const float *slots, const float *org, const float *dir, size_t n, float tnear, float *t, int32_t *prim,
                   float *u, float *v, int nthreads, unsigned long long *steps) {
  Job J;
  memset(&J, 0, sizeof J);
  J.nodes4 = nodes4; J.slots = slots; J.has_nodes = n_nodes4 > 0; J.org = org; J.dir = dir; J.n = n; J.tnear = tnear; J.any = 0; J.t = t; J.prim = prim; J.u = u; J.v = v;
  return run(&J, nthreads, steps);
}
/* any hit (rtcOccluded): occ[j] = 1 when ray j meets a triangle beyond tnear (no upper bound, like EmbreeMeshAdapter.cpp:375) */
int simd_occluded(const uint32_t *nodes4, size_t n_nodes4, const float *slots, const float *org, const float *dir, size_t n, float tnear, uint8_t *occ, int nthreads,
                  unsigned long long *steps) {
  Job J;
  memset(&J, 0, sizeof J);
  J.nodes4 = nodes4; J.slots = slots; J.has_nodes = n_nodes4 > 0; J.org = org; J.dir = dir; J.n = n; J.tnear = tnear; J.any = 1; J.occ = occ;
  return run(&J, nthreads, steps);
}
