/* simd8_baseline.c -- the 8-WIDE SIMD CPU baseline of bench.py's `cpu_baseline` leg (AVX2 + FMA), beside the 4-wide SSE one (simd_baseline.c).  MEASUREMENT
 * INFRASTRUCTURE, like the rest of oracle/: nothing under gravit_amd/ links, loads or calls it; it is not the oracle -- its hits are validated against the
 * oracle's before its time is reported (bench.py, tests/test_gpu_parity.py).
 *
 * Why: the reference picks its packet / node width from the host ISA at build time (GVT_AVX_TARGET / GVT_AVX2_TARGET / GVT_AVX512KNL_TARGET select
 * rtcIntersect8 / 16 and the 8 / 16-wide ray packets, src/gvt/render/adapter/embree/EmbreeMeshAdapter.cpp:50-74, :474); on an AVX2 host Embree 2.x walks a BVH8,
 * one ray against EIGHT child boxes per step.  This file does that over the same tree the GPU built: simd8_build collapses the downloaded compressed 4-wide nodes
 * once more on the host -- the inner child of largest area whose own children still fit is replaced by them until none fits (the greedy rule of the device's own
 * collapse, csrc/lbvh.hip k_collapse4, carried on to eight slots) -- into 256-byte nodes of eight float boxes (the 8-bit planes decoded, every plane moved outward by one ulp), and
 * simd8_intersect / simd8_occluded walk them with six 8-wide fused multiply-adds per step, nearest child first, a stack per ray; leaves of <= 2 triangles go through
 * the restated Moeller-Trumbore test of the kernels (strict IEEE, no contraction: the same (t, primID, u, v) bits -- boxes are conservative, they cannot change a
 * result).  The build is not timed (neither is the GPU's, nor Embree's rtcCommit in the reference's own timers).
 * gcc -O3 -mavx2 -mfma -ffp-contract=off (oracle/Makefile); simd8_supported() says whether the host can run it. */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SB_FLT_MAX 3.402823466e+38f
#define SB_STACK 512
#define SB_CHUNK 4096

typedef struct { float lo[3][8], hi[3][8]; int32_t ref[8]; int32_t n, pad[7]; } Node8; /* 256 bytes */

int simd8_supported(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

/* ---- the 8-wide collapse of the 4-wide tree ---- */
typedef struct { float lo[3], hi[3]; int32_t ref; } Slot;
static inline float area_of(const Slot *s) { const float dx = s->hi[0] - s->lo[0], dy = s->hi[1] - s->lo[1], dz = s->hi[2] - s->lo[2]; return dx * dy + dy * dz + dz * dx; }
/* children of compressed node `nd` (include/gvt_hip.h, gvt_hip_mesh_download_wide: w0 = origin xyz | step x, w1 = x lo / x hi / y lo / y hi planes, w2 = z lo / z hi planes |
 * ref0 ref1, w3 = ref2 ref3 | step y | step z; an unused slot has lo plane 255 and hi plane 0) */
static int children_of(const uint32_t *nd, Slot *out) {
  float org[3], step[3];
  memcpy(org, nd, 12);
  memcpy(&step[0], nd + 3, 4); memcpy(&step[1], nd + 14, 4); memcpy(&step[2], nd + 15, 4);
  const int32_t refs[4] = { (int32_t)nd[10], (int32_t)nd[11], (int32_t)nd[12], (int32_t)nd[13] };
  int n = 0;
  for (int c = 0; c < 4; c++) {
    Slot s;
    int empty = 0;
    for (int a = 0; a < 3; a++) {
      const uint32_t ql = (nd[4 + 2 * a] >> (8 * c)) & 0xffu, qh = (nd[5 + 2 * a] >> (8 * c)) & 0xffu;
      if (ql > qh) empty = 1;
      s.lo[a] = nextafterf(fmaf((float)ql, step[a], org[a]), -INFINITY);
      s.hi[a] = nextafterf(fmaf((float)qh, step[a], org[a]), INFINITY);
    }
    if (empty) continue;
    s.ref = refs[c];
    out[n++] = s;
  }
  return n;
}
/* nodes8 must hold n_nodes4 entries (an 8-wide node is made per 4-wide node that stays a root: never more); returns the number made, < 0 on failure */
long simd8_build(const uint32_t *nodes4, size_t n_nodes4, Node8 *nodes8) {
  if (!n_nodes4) return 0;
  int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * n_nodes4); /* 4-wide root of 8-wide node k, breadth first */
  if (!queue) return -1;
  size_t head = 0, tail = 0;
  queue[tail++] = 0;
  while (head < tail) {
    const uint32_t *nd = nodes4 + 16 * (size_t)queue[head];
    Slot s[8], tmp[4];
    int n = children_of(nd, s);
    for (;;) { /* open the inner child of largest area whose children still fit, until none does */
      int k = -1, m = 0;
      float best = -1.f;
      for (int c = 0; c < n; c++) {
        if (s[c].ref < 0) continue;
        const float a = area_of(&s[c]);
        if (a <= best) continue;
        Slot probe[4];
        const int mc = children_of(nodes4 + 16 * (size_t)s[c].ref, probe);
        if (mc == 0 || n - 1 + mc > 8) continue;
        best = a; k = c; m = mc;
        memcpy(tmp, probe, sizeof probe);
      }
      if (k < 0) break;
      s[k] = tmp[0];
      for (int c = 1; c < m; c++) s[n++] = tmp[c];
    }
    Node8 *o = nodes8 + head;
    memset(o, 0, sizeof *o);
    o->n = n;
    for (int c = 0; c < 8; c++) {
      if (c < n) {
        for (int a = 0; a < 3; a++) { o->lo[a][c] = s[c].lo[a]; o->hi[a][c] = s[c].hi[a]; }
        if (s[c].ref >= 0) { if (tail == n_nodes4) { free(queue); return -2; } queue[tail] = s[c].ref; o->ref[c] = (int32_t)tail; tail++; }
        else o->ref[c] = s[c].ref;
      } else {
        for (int a = 0; a < 3; a++) { o->lo[a][c] = SB_FLT_MAX; o->hi[a][c] = -SB_FLT_MAX; } /* an inverted box no ray enters */
        o->ref[c] = INT32_MIN;
      }
    }
    head++;
  }
  free(queue);
  return (long)tail;
}

/* ---- traversal ---- */
typedef struct {
  const Node8 *nodes;
  const float *slots;
  int has_nodes;
  const float *org, *dir;
  size_t n;
  float tnear;
  int any;
  float *t, *u, *v;
  int32_t *prim;
  uint8_t *occ;
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
static inline float sb_cull(float bt) { return fmaf(fabsf(bt), 0x1p-10f, bt); }

static void trace_one(const Job *J, size_t j, unsigned long long *n_node, unsigned long long *n_leaf) {
  const V3 O = mk3(J->org[3 * j], J->org[3 * j + 1], J->org[3 * j + 2]), D = mk3(J->dir[3 * j], J->dir[3 * j + 1], J->dir[3 * j + 2]);
  const float dx = fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x, dy = fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y,
              dz = fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z;
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
  const float ox = O.x * ix, oy = O.y * iy, oz = O.z * iz;
  /* slabs widened like the kernels' RaySlab: near offsets moved out, far offsets in, by 2^-21 |O / d|; far distances x (1 + 2^-21) */
  const float ex = fabsf(ox) * 4.76837158e-7f, ey = fabsf(oy) * 4.76837158e-7f, ez = fabsf(oz) * 4.76837158e-7f;
  const __m256 vix = _mm256_set1_ps(ix), viy = _mm256_set1_ps(iy), viz = _mm256_set1_ps(iz);
  const __m256 oxn = _mm256_set1_ps(-(ox + ex)), oxf = _mm256_set1_ps(-(ox - ex)), oyn = _mm256_set1_ps(-(oy + ey)), oyf = _mm256_set1_ps(-(oy - ey)),
               ozn = _mm256_set1_ps(-(oz + ez)), ozf = _mm256_set1_ps(-(oz - ez));
  /* near / far plane arrays by the ray's direction sign: offsets into Node8 (lo[a] at a * 8 floats, hi[a] at (3 + a) * 8) */
  const int nxo = ix >= 0.f ? 0 : 24, fxo = ix >= 0.f ? 24 : 0, nyo = iy >= 0.f ? 8 : 32, fyo = iy >= 0.f ? 32 : 8, nzo = iz >= 0.f ? 16 : 40, fzo = iz >= 0.f ? 40 : 16;
  float bt = SB_FLT_MAX, bu = 0.f, bv = 0.f, bden = 1.f;
  int bp = -1;
  int32_t st_ref[SB_STACK];
  float st_tn[SB_STACK];
  int sp = 0;
  int32_t cur = J->has_nodes ? 0 : INT32_MIN;
  int done = !J->has_nodes;
#define SB_POP() { cur = INT32_MIN; while (sp) { sp--; if (J->any || st_tn[sp] <= sb_cull(bt)) { cur = st_ref[sp]; break; } } if (cur == INT32_MIN) break; }
  while (!done) {
    if (cur >= 0) {
      (*n_node)++;
      const Node8 *nd = J->nodes + cur;
      const float *f = (const float *)nd;
      const __m256 nx = _mm256_fmadd_ps(_mm256_load_ps(f + nxo), vix, oxn), ny = _mm256_fmadd_ps(_mm256_load_ps(f + nyo), viy, oyn), nz = _mm256_fmadd_ps(_mm256_load_ps(f + nzo), viz, ozn);
      const __m256 fx = _mm256_fmadd_ps(_mm256_load_ps(f + fxo), vix, oxf), fy = _mm256_fmadd_ps(_mm256_load_ps(f + fyo), viy, oyf), fz = _mm256_fmadd_ps(_mm256_load_ps(f + fzo), viz, ozf);
      const __m256 tnr = _mm256_max_ps(_mm256_max_ps(nx, ny), _mm256_max_ps(nz, _mm256_setzero_ps()));
      const __m256 tfr = _mm256_mul_ps(_mm256_min_ps(_mm256_min_ps(fx, fy), fz), _mm256_set1_ps(1.0000004f));
      const __m256 lim = _mm256_set1_ps(J->any ? SB_FLT_MAX : sb_cull(bt));
      int mask = _mm256_movemask_ps(_mm256_cmp_ps(tnr, _mm256_min_ps(tfr, lim), _CMP_LE_OQ));
      if (!mask) { SB_POP(); continue; }
      float tn[8] __attribute__((aligned(32)));
      _mm256_store_ps(tn, tnr);
      int idx[8], k = 0; /* the entered children nearest first (insertion into a list of at most eight) */
      while (mask) {
        const int c = __builtin_ctz((unsigned)mask);
        mask &= mask - 1;
        int p = k++;
        while (p > 0 && tn[idx[p - 1]] > tn[c]) { idx[p] = idx[p - 1]; p--; }
        idx[p] = c;
      }
      for (int p = k - 1; p >= 1; p--) {
        if (sp == SB_STACK) { atomic_fetch_add(&((Job *)J)->overflow, 1ull); done = 1; break; }
        st_ref[sp] = nd->ref[idx[p]]; st_tn[sp] = tn[idx[p]]; sp++;
      }
      cur = nd->ref[idx[0]];
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
static int run(Job *J, int nthreads, unsigned long long *steps) {
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
/* nodes8: 32-byte aligned (the binding allocates it so) */
int simd8_intersect(const Node8 *nodes8, size_t n_nodes8, const float *slots, const float *org, const float *dir, size_t n, float tnear, float *t, int32_t *prim,
                    float *u, float *v, int nthreads, unsigned long long *steps) {
  Job J;
  memset(&J, 0, sizeof J);
  J.nodes = nodes8; J.slots = slots; J.has_nodes = n_nodes8 > 0; J.org = org; J.dir = dir; J.n = n; J.tnear = tnear; J.any = 0; J.t = t; J.prim = prim; J.u = u; J.v = v;
  return run(&J, nthreads, steps);
}
int simd8_occluded(const Node8 *nodes8, size_t n_nodes8, const float *slots, const float *org, const float *dir, size_t n, float tnear, uint8_t *occ, int nthreads,
                   unsigned long long *steps) {
  Job J;
  memset(&J, 0, sizeof J);
  J.nodes = nodes8; J.slots = slots; J.has_nodes = n_nodes8 > 0; J.org = org; J.dir = dir; J.n = n; J.tnear = tnear; J.any = 1; J.occ = occ;
  return run(&J, nthreads, steps);
}
