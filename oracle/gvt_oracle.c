/*
 * gvt_oracle.c -- CPU restatement of the GraviT engine-adapter hot path (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY (see gvt_oracle.h).  Compile with -ffp-contract=off and
 * without -ffast-math: every float expression below is written in the evaluation
 * order of the reference source it cites (glm 0.9.8.1 order for vector ops), so that
 * the HIP kernels, written in the same order, can be compared bit for bit.
 *
 * Citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#define _GNU_SOURCE /* sched_getaffinity, pthread_setaffinity_np: the worker pool pins its threads */
#include "gvt_oracle.h"
#include "../include/gvt_math.h" /* acos / sinf / cosf of the bounce path: the definition shared with the device code */

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RAY_EPSILON 1.e-6f /* actor/Ray.cpp:33 */

/* ------------------------------------------------------------------------- */
/* glm-ordered vector helpers (third-party/glm/glm/detail/func_geometric.inl) */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* compute_dot<tvec3>: tmp = x*y; tmp.x + tmp.y + tmp.z   (func_geometric.inl:54-61) */
static inline float dot3(v3 a, v3 b) { v3 t = mul3(a, b); return t.x + t.y + t.z; }
/* compute_cross (func_geometric.inl:74-85) */
static inline v3 cross3(v3 x, v3 y) {
  return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
/* compute_normalize: v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt (func_exponential.inl:130-133) */
static inline v3 norm3(v3 a) { return scl3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float fmin_ref(float a, float b) { return (a < b) ? a : b; } /* RayPacket.h:51 */
static inline float fmax_ref(float a, float b) { return (a > b) ? a : b; } /* RayPacket.h:59 */

/* glm mat4 (column-major, m[c*4+r]) * vec4: (m0*v0 + m1*v1) + (m2*v2 + m3*v3)  (type_mat4x4.inl operator*) */
static inline v3 xfm_point(const float *m, v3 p) {
  return V3((m[0] * p.x + m[4] * p.y) + (m[8] * p.z + m[12] * 1.0f), (m[1] * p.x + m[5] * p.y) + (m[9] * p.z + m[13] * 1.0f),
            (m[2] * p.x + m[6] * p.y) + (m[10] * p.z + m[14] * 1.0f));
}
static inline v3 xfm_vector(const float *m, v3 d) {
  return V3((m[0] * d.x + m[4] * d.y) + (m[8] * d.z + m[12] * 0.0f), (m[1] * d.x + m[5] * d.y) + (m[9] * d.z + m[13] * 0.0f),
            (m[2] * d.x + m[6] * d.y) + (m[10] * d.z + m[14] * 0.0f));
}
/* glm mat3 (n[c*3+r]) * vec3: m00*x + m10*y + m20*z, left to right (type_mat3x3.inl operator*) */
static inline v3 mat3_mul(const float *n, v3 v) {
  return V3(n[0] * v.x + n[3] * v.y + n[6] * v.z, n[1] * v.x + n[4] * v.y + n[7] * v.z, n[2] * v.x + n[5] * v.y + n[8] * v.z);
}

/* ------------------------------------------------------------------------- */
/* RandEngine (core/math/RandEngine.h:43-81)                                  */
/* ------------------------------------------------------------------------- */
#define ROTL32(r, n) (((r) << (n)) | ((r) >> (32 - (n))))
float orc_rng(uint32_t *seed) {
  uint32_t x, y, z;
  x = (*seed >> 16) + 4125832013u;
  y = (*seed & 0xffff) + 814584116u;
  z = 542;
  x *= 255519323u;
  x = ROTL32(x, 13);
  y *= 3166389663u;
  y = ROTL32(y, 17);
  z -= ROTL32(z, 11);
  z = ROTL32(z, 27);
  *seed = x ^ y ^ z;
  return ((float)(*seed & 0x00FFFFFF) / (float)0x01000000);
}
static inline float fastrand01(uint32_t *seed) { return 0.0f + orc_rng(seed) * (1.0f - 0.0f); } /* :73-76 */
float orc_fastrand_lcg(uint32_t *seedval, float mn, float mx) { /* :78-81 */
  const float ff = (1.0f / 65535.0f);
  *seedval = 214013u * (*seedval) + 2531011u;
  return mn + (*seedval >> 16) * ff * (mx - mn);
}
/* per-ray stream seed: the reference seeds one engine per TBB chunk with the chunk's first ray index
 * (EmbreeMeshAdapter.cpp:446-447), which is schedule dependent; the restatement gives every ray its own stream so that any
 * execution order, list order and rank count give the same values.  A stream starts either from (call seed, index of the ray in
 * rayList) -- Adapter::trace on a host RayVector, orc_trace -- or from the word the ray carries in bytes 64..67 of its 80-byte
 * image (inside Ray::data[68]: copied by every Ray copy / pack of the reference, never read by it, actor/Ray.h:95,128-151), set by
 * the camera and advanced by every draw -- the schedulers, orc_render_image / orc_render_domain.  Word 0 = no stream yet. */
static inline uint32_t ray_stream_seed(uint32_t seed, uint64_t index) {
  uint32_t s = seed ^ (uint32_t)(index * 0x9E3779B9u) ^ (uint32_t)(index >> 32);
  s ^= s >> 16; s *= 0x85EBCA6Bu; s ^= s >> 13; s *= 0xC2B2AE35u; s ^= s >> 16;
  return s;
}
static inline uint32_t camera_stream_word(uint64_t ridx) { /* ridx = position in generateRays' list (gvtCamera.cpp:262-305) */
  const uint32_t s = ray_stream_seed(0x243F6A88u, ridx);
  return s ? s : 0x9E3779B9u;
}
static inline uint32_t ray_word(const orc_ray *r) { uint32_t w; memcpy(&w, &r->pad[0], 4); return w; }
static inline void set_ray_word(orc_ray *r, uint32_t w) { memcpy(&r->pad[0], &w, 4); }

/* ---- known misses (NOT in the reference; an image-identical shortcut of the build's schedulers, restated here so that the checker
 * can follow it ray for ray).  shuffleRays (TracerBase.h:392-400) sends a ray that left instance A without a hit to the nearest other
 * instance box ahead of it, origin advanced by 95 % of the distance (:393).  Where the boxes of A and B overlap, a ray that has
 * crossed A and B without a hit is handed back and forth -- A, B, A, B, ... five or six times, each hop a full traversal that CANNOT
 * find anything: the ray is the same half-line from an origin further along, and the instance held nothing on the longer one.
 * With the shortcut on, a ray carries the instances it has already crossed without a hit on its current straight segment (six 16-bit
 * entries, instance + 1, in bytes 68..79 of the 80-byte Ray image: inside Ray::data, copied by every copy / pack of the reference and
 * never read by it, actor/Ray.h:95,128-151); when shuffleRays' choice is such an instance, the trace there is taken as the miss it
 * must be: the origin advance is replayed with the reference's arithmetic and the next choice is made as if the ray came from there.
 * A bounce starts a new segment (list cleared); a shadow ray is born with an empty list.  The ray that finally reaches a queue, or the
 * framebuffer, is bit for bit the ray the reference's hops would have delivered; only the rays_closest / adapter-call / rays_sent
 * counts drop.  Off (the default here): the reference's behaviour, hop by hop. */
static int g_skip_known = 0;
void orc_set_skip_known_misses(int on) { g_skip_known = on ? 1 : 0; }
int orc_get_skip_known_misses(void) { return g_skip_known; }
static inline void km_get(const orc_ray *r, uint16_t e[6]) { memcpy(e, &r->pad[1], 12); }
static inline void km_put(orc_ray *r, const uint16_t e[6]) { memcpy(&r->pad[1], e, 12); }
static inline void km_clear(orc_ray *r) { memset(&r->pad[1], 0, 12); }
static inline int km_has(const uint16_t e[6], int inst) {
  if (inst < 0 || inst >= 65535) return 0;
  const uint16_t v = (uint16_t)(inst + 1);
  return e[0] == v || e[1] == v || e[2] == v || e[3] == v || e[4] == v || e[5] == v;
}
static inline void km_add(uint16_t e[6], int inst) { /* first free entry; a full list forgets its oldest entry (forgetting only costs a trace) */
  if (inst < 0 || inst >= 65535 || km_has(e, inst)) return;
  const uint16_t v = (uint16_t)(inst + 1);
  for (int k = 0; k < 6; k++) if (!e[k]) { e[k] = v; return; }
  for (int k = 0; k < 5; k++) e[k] = e[k + 1];
  e[5] = v;
}

/* ------------------------------------------------------------------------- */
/* Lights (data/scene/Light.cpp:58-133)                                       */
/* ------------------------------------------------------------------------- */
void orc_light_contribution(const orc_light *L, const float hit[3], const float samplePos[3], float out[3]) {
  v3 c = ld3(L->color);
  if (L->type == ORC_LIGHT_AMBIENT) { st3(out, c); return; } /* :70 */
  v3 p = (L->type == ORC_LIGHT_AREA) ? ld3(samplePos) : ld3(L->position); /* :58-62 / :129-133 */
  float distance = 1.f / len3(sub3(p, ld3(hit)));
  distance = (distance > 1.f) ? 1.f : distance;
  st3(out, scl3(c, distance));
}

/* AreaLight ctor basis (:72-99) and GetPosition (:115-127) */
static void area_light_basis(const orc_light *L, v3 *u, v3 *w) {
  v3 v = ld3(L->normal), up = V3(0, 1, 0);
  if (v.x == up.x && v.y == up.y && v.z == up.z) {
    *u = V3(1, 0, 0);
    *w = V3(0, 0, 1);
  } else {
    u->x = up.y * v.z - v.y * up.z; u->y = up.z * v.x - v.z * up.x; u->z = up.x * v.y - v.x * up.y;
    w->x = v.y * u->z - u->y * v.z; w->y = v.z * u->x - u->z * v.x; w->z = v.x * u->y - u->x * v.y;
  }
}
static v3 area_light_position(const orc_light *L, uint32_t *seed) {
  v3 u, w;
  area_light_basis(L, &u, &w);
  float xLocation = (float)((orc_fastrand_lcg(seed, 0, 1) - 0.5) * L->width);
  float zLocation = (float)((orc_fastrand_lcg(seed, 0, 1) - 0.5) * L->height);
  float xCoord = xLocation * u.x + zLocation * w.x;
  float yCoord = xLocation * u.y + zLocation * w.y;
  float zCoord = xLocation * u.z + zLocation * w.z;
  return V3(L->position[0] + xCoord, L->position[1] + yCoord, L->position[2] + zCoord);
}

/* ------------------------------------------------------------------------- */
/* Shade (data/primitives/Material.cpp:50-139)                                */
/* ------------------------------------------------------------------------- */
static float clamp01(float x) { const float a = (x < 0.f) ? 0.f : x; return (1.f < a) ? 1.f : a; } /* embree clamp(x) = min(max(x,0),1) */
int orc_shade(const orc_material *mat, const orc_ray *ray, const float Nf[3], const orc_light *light,
              const float lightPos[3], float color_out[3]) {
  v3 N = ld3(Nf), o = ld3(ray->origin), d = ld3(ray->direction);
  v3 hitPoint = add3(o, scl3(d, ray->t));
  v3 wi = norm3(sub3(ld3(lightPos), hitPoint));
  float dNw = dot3(N, wi);
  float NdotL = (0.f < dNw) ? dNw : 0.f; /* std::max(0.f, x) */
  float Li[3], hp[3];
  st3(hp, hitPoint);
  orc_light_contribution(light, hp, lightPos, Li);
  if (NdotL == 0.f || (Li[0] == 0.f && Li[1] == 0.f && Li[2] == 0.f)) return 0;

  v3 kd = ld3(mat->kd), ks = ld3(mat->ks), color;
  switch (mat->type) {
  case 0: /* lambertShade :50-57 */
    color = scl3(kd, NdotL * ray->w);
    break;
  case 1: { /* phongShade :59-70 */
    v3 R = sub3(scl3(scl3(N, 2.f), NdotL), wi);
    float vr = dot3(R, neg3(d));
    float VdotR = (0.f < vr) ? vr : 0.f;
    float power = VdotR * powf(VdotR, mat->alpha);
    color = scl3(kd, NdotL * ray->w);
    color = add3(color, scl3(ks, power * ray->w));
  } break;
  case 2: { /* blinnPhongShade :72-87 */
    v3 H = norm3(sub3(wi, d));
    float hn = dot3(H, N);
    float NdotH = (0.f < hn) ? hn : 0.f;
    float power = NdotH * powf(NdotH, mat->alpha);
    v3 diffuse = scl3(kd, NdotL * ray->w);
    v3 specular = scl3(ks, power * ray->w);
    color = add3(diffuse, specular);
  } break;
  case 3:   /* EMBREE_MATERIAL_METAL  */
  case 4:   /* EMBREE_MATERIAL_VELVET */
  case 5: { /* EMBREE_MATERIAL_MATTE  -- Material.cpp:106-122 -> Material__eval (adapter/embree/EmbreeMaterial.h:289-314) */
    /* dg.Ns = surfaceNormal, wo = -ray.direction; Embree's rcp()/rsqrt() (SSE estimate + one Newton step,
       embree-shaders/common/math/math.h:59-85, vec3fa.h:124-144) are restated as 1/x and 1/sqrt(x): parity with the reference build
       is to ~1e-6 relative for these three types, not to the bit */
    v3 wo = neg3(d), r = V3(0, 0, 0);
    const float one_over_pi = 0.31830988618379069122f;
    if (mat->type == 5) { /* MatteMaterial__eval :168-172: Lambertian R * clamp(dot(wi, Ns)) */
      r = scl3(kd, clamp01(dot3(wi, N)));
    } else if (mat->type == 4) { /* VelvetMaterial__eval :246-253 = Minneart (:188-192) + Velvety (:222-228) */
      const float cosThetaI = clamp01(dot3(wi, N));
      const float backScatter = powf(clamp01(dot3(wo, wi)), mat->backScattering);
      v3 a = scl3(ks, backScatter * cosThetaI * one_over_pi);
      const float cosThetaO = clamp01(dot3(wo, N));
      const float sinThetaO = sqrtf(1.0f - cosThetaO * cosThetaO);
      const float horizonScatter = powf(sinThetaO, mat->horizonScatteringFallOff);
      v3 b = scl3(ld3(mat->horizonScatteringColor), horizonScatter * cosThetaI * one_over_pi);
      r = add3(a, b);
    } else { /* MetalMaterial__eval :259-277, optics.h:75-83 (fresnelConductor), :131-137 (PowerCosineDistribution) */
      const float expo = 1.0f / mat->roughness;
      const float cosThetaO = dot3(wo, N), cosThetaI = dot3(wi, N);
      if (!(cosThetaI <= 0.0f || cosThetaO <= 0.0f)) {
        v3 s_ = add3(wi, wo);
        v3 wh = scl3(s_, 1.0f / sqrtf(dot3(s_, s_)));
        const float cosThetaH = dot3(wh, N);
        const float cosTheta = dot3(wi, wh);
        v3 eta = ld3(mat->eta), k = ld3(mat->k), F;
        {
          const float cosi = cosTheta, c2 = cosi * cosi;
          v3 tmp = add3(mul3(eta, eta), mul3(k, k));
          v3 two_eta_c = scl3(scl3(eta, 2.0f), cosi);
          v3 one = V3(1.0f, 1.0f, 1.0f), vc2 = V3(c2, c2, c2);
          v3 num1 = add3(sub3(scl3(tmp, c2), two_eta_c), one), den1 = add3(add3(scl3(tmp, c2), two_eta_c), one);
          v3 num2 = add3(sub3(tmp, two_eta_c), vc2), den2 = add3(add3(tmp, two_eta_c), vc2);
          v3 Rpar = V3(num1.x / den1.x, num1.y / den1.y, num1.z / den1.z);
          v3 Rper = V3(num2.x / den2.x, num2.y / den2.y, num2.z / den2.z);
          F = scl3(add3(Rpar, Rper), 0.5f);
        }
        const float D = (expo + 2) * (1.0f / (2.0f * 3.14159265358979323846f)) * powf(fabsf(cosThetaH), expo);
        const float g1 = 2.0f * cosThetaH * cosThetaO / cosTheta, g2 = 2.0f * cosThetaH * cosThetaI / cosTheta;
        const float gm = (g1 < g2) ? g1 : g2;
        const float G = (1.0f < gm) ? 1.0f : gm;
        r = scl3(scl3(scl3(mul3(ks, F), D), G), 1.0f / (4.0f * cosThetaO));
      }
    }
    color = scl3(scl3(r, 2.f), ray->w); /* 2.f * glm::vec3(r) * ray.w :120 */
  } break;
  default: /* :128-131: prints and leaves `color` untouched; restated as black */
    color = V3(0, 0, 0);
    break;
  }
  color = mul3(color, ld3(Li)); /* :134 */
  /* glm::clamp(c,0,1) = min(max(c,0),1) */
  float c[3] = { color.x, color.y, color.z };
  for (int i = 0; i < 3; i++) {
    float a = (c[i] < 0.f) ? 0.f : c[i];
    color_out[i] = (1.f < a) ? 1.f : a;
  }
  return 1;
}

/* ------------------------------------------------------------------------- */
/* Mesh + acceleration structure                                              */
/* ------------------------------------------------------------------------- */
typedef struct {
  float lo[3], hi[3];
  int32_t left;  /* inner: index of left child (right = left+1); leaf: first prim slot */
  int32_t count; /* 0 = inner, >0 = leaf prim count */
} orc_node;

struct orc_mesh {
  size_t nV, nT;
  float *verts;     /* nV*3 */
  int32_t *tris;    /* nT*3 */
  float *normals;   /* nV*3 */
  float *vcolors;   /* nV*3 or NULL */
  orc_material *materials;
  size_t nMat;
  int32_t *face_mat; /* nT or NULL */
  orc_material mesh_mat;
  /* BVH */
  orc_node *nodes;
  size_t nNodes;
  int32_t *prim_idx; /* nT, leaf order */
  float lo[3], hi[3];
};

void orc_generate_normals(const float *verts, size_t nV, const int32_t *tris, size_t nT, float *normals) {
  /* Mesh::generateNormals, Mesh.cpp:116-154: unweighted sum of normalized face normals */
  for (size_t i = 0; i < nV * 3; i++) normals[i] = 0.0f;
  for (size_t i = 0; i < nT; i++) {
    int I = tris[3 * i], J = tris[3 * i + 1], K = tris[3 * i + 2];
    v3 a = ld3(verts + 3 * I), b = ld3(verts + 3 * J), c = ld3(verts + 3 * K);
    v3 u = sub3(b, a), v = sub3(c, a), n;
    n.x = u.y * v.z - u.z * v.y;
    n.y = u.z * v.x - u.x * v.z;
    n.z = u.x * v.y - u.y * v.x;
    n = norm3(n);
    st3(normals + 3 * I, add3(ld3(normals + 3 * I), n));
    st3(normals + 3 * J, add3(ld3(normals + 3 * J), n));
    st3(normals + 3 * K, add3(ld3(normals + 3 * K), n));
  }
  for (size_t i = 0; i < nV; i++) st3(normals + 3 * i, norm3(ld3(normals + 3 * i)));
}

static void default_material(orc_material *m) { /* Material.h:62-77 */
  memset(m, 0, sizeof *m);
  m->type = 0;
  m->kd[0] = m->kd[1] = m->kd[2] = .5f;
  m->ks[0] = m->ks[1] = m->ks[2] = .5f;
  m->alpha = 1.f;
  m->eta[0] = .19f; m->eta[1] = 1.45f; m->eta[2] = 1.50f;
  m->k[0] = 3.06f; m->k[1] = 2.40f; m->k[2] = 1.88f;
  m->roughness = 0.05f;
}

/* --- BVH build: spatial-median split on the centroid box, leaf <= 4 (the oracle's own choice: Embree's
 *     builder is not in the tree; any conservative structure returns the same closest hit) --- */
typedef struct {
  const float *clo, *chi; /* per-prim bounds */
  float *cen;             /* per-prim centroid */
  int32_t *idx;
  orc_node *nodes;
  size_t nNodes, capNodes;
  float pad;
} bvh_builder;

#define ORC_LEAF 4

static void node_bounds(bvh_builder *B, size_t a, size_t b, float lo[3], float hi[3], float clo[3], float chi[3]) {
  for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; clo[k] = FLT_MAX; chi[k] = -FLT_MAX; }
  for (size_t i = a; i < b; i++) {
    int32_t p = B->idx[i];
    for (int k = 0; k < 3; k++) {
      float l = B->clo[3 * p + k], h = B->chi[3 * p + k], c = B->cen[3 * p + k];
      if (l < lo[k]) lo[k] = l;
      if (h > hi[k]) hi[k] = h;
      if (c < clo[k]) clo[k] = c;
      if (c > chi[k]) chi[k] = c;
    }
  }
}

static void build_rec(bvh_builder *B, int32_t ni, size_t a, size_t b) {
  float lo[3], hi[3], clo[3], chi[3];
  node_bounds(B, a, b, lo, hi, clo, chi);
  orc_node *n = &B->nodes[ni];
  for (int k = 0; k < 3; k++) { n->lo[k] = lo[k] - B->pad; n->hi[k] = hi[k] + B->pad; }
  if (b - a <= ORC_LEAF) { n->left = (int32_t)a; n->count = (int32_t)(b - a); return; }
  int ax = 0;
  if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
  if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
  float mid = 0.5f * (clo[ax] + chi[ax]);
  size_t i = a, j = b;
  while (i < j) {
    if (B->cen[3 * B->idx[i] + ax] < mid) i++;
    else { j--; int32_t t = B->idx[i]; B->idx[i] = B->idx[j]; B->idx[j] = t; }
  }
  if (i == a || i == b) i = a + (b - a) / 2; /* degenerate: split the index range in half */
  int32_t l = (int32_t)B->nNodes;
  B->nNodes += 2;
  n = &B->nodes[ni];
  n->left = l;
  n->count = 0;
  build_rec(B, l, a, i);
  build_rec(B, l + 1, i, b);
}

static void build_bvh(orc_mesh *M) {
  size_t nT = M->nT;
  float *clo = malloc(sizeof(float) * 3 * (nT + 1)), *chi = malloc(sizeof(float) * 3 * (nT + 1));
  float *cen = malloc(sizeof(float) * 3 * (nT + 1));
  M->prim_idx = malloc(sizeof(int32_t) * (nT + 1));
  for (int k = 0; k < 3; k++) { M->lo[k] = FLT_MAX; M->hi[k] = -FLT_MAX; }
  for (size_t i = 0; i < nT; i++) {
    M->prim_idx[i] = (int32_t)i;
    for (int k = 0; k < 3; k++) {
      float a = M->verts[3 * M->tris[3 * i] + k], b = M->verts[3 * M->tris[3 * i + 1] + k],
            c = M->verts[3 * M->tris[3 * i + 2] + k];
      float l = fminf(a, fminf(b, c)), h = fmaxf(a, fmaxf(b, c));
      clo[3 * i + k] = l; chi[3 * i + k] = h; cen[3 * i + k] = 0.5f * (l + h);
      if (l < M->lo[k]) M->lo[k] = l;
      if (h > M->hi[k]) M->hi[k] = h;
    }
  }
  bvh_builder B = { clo, chi, cen, M->prim_idx, NULL, 0, 0, 0.f };
  B.capNodes = 2 * nT + 2;
  B.nodes = malloc(sizeof(orc_node) * B.capNodes);
  /* outward pad: keeps the slab test conservative w.r.t. the triangle test's rounding */
  float ext = 0.f;
  for (int k = 0; k < 3; k++) { ext = fmaxf(ext, fabsf(M->lo[k])); ext = fmaxf(ext, fabsf(M->hi[k])); }
  B.pad = (nT ? ext : 0.f) * 1e-5f;
  B.nNodes = 1;
  if (nT) build_rec(&B, 0, 0, nT);
  else { memset(&B.nodes[0], 0, sizeof(orc_node)); B.nodes[0].count = 0; B.nodes[0].left = -1; }
  M->nodes = B.nodes;
  M->nNodes = B.nNodes;
  free(clo); free(chi); free(cen);
}

orc_mesh *orc_mesh_create(const float *verts, size_t nV, const int32_t *tris, size_t nT, const float *vnormals,
                          const float *vcolors, const orc_material *materials, size_t nMat, const int32_t *face_mat,
                          const orc_material *mesh_mat) {
  orc_mesh *M = calloc(1, sizeof *M);
  M->nV = nV; M->nT = nT;
  M->verts = malloc(sizeof(float) * 3 * (nV + 1));
  memcpy(M->verts, verts, sizeof(float) * 3 * nV);
  M->tris = malloc(sizeof(int32_t) * 3 * (nT + 1));
  memcpy(M->tris, tris, sizeof(int32_t) * 3 * nT);
  M->normals = malloc(sizeof(float) * 3 * (nV + 1));
  if (vnormals) memcpy(M->normals, vnormals, sizeof(float) * 3 * nV);
  else orc_generate_normals(M->verts, nV, M->tris, nT, M->normals); /* EmbreeMeshAdapter.cpp:129 */
  if (vcolors) { M->vcolors = malloc(sizeof(float) * 3 * nV); memcpy(M->vcolors, vcolors, sizeof(float) * 3 * nV); }
  if (materials && nMat) {
    M->materials = malloc(sizeof(orc_material) * nMat);
    memcpy(M->materials, materials, sizeof(orc_material) * nMat);
    M->nMat = nMat;
  }
  if (face_mat) { M->face_mat = malloc(sizeof(int32_t) * (nT + 1)); memcpy(M->face_mat, face_mat, sizeof(int32_t) * nT); }
  if (mesh_mat) M->mesh_mat = *mesh_mat; else default_material(&M->mesh_mat);
  build_bvh(M);
  return M;
}

void orc_mesh_destroy(orc_mesh *M) {
  if (!M) return;
  free(M->verts); free(M->tris); free(M->normals); free(M->vcolors); free(M->materials); free(M->face_mat);
  free(M->nodes); free(M->prim_idx); free(M);
}
const float *orc_mesh_normals(const orc_mesh *M) { return M->normals; }
void orc_mesh_bbox(const orc_mesh *M, float lo[3], float hi[3]) { memcpy(lo, M->lo, 12); memcpy(hi, M->hi, 12); }

/* ------------------------------------------------------------------------- */
/* Triangle test.  Restates Embree 2.x's Moeller-Trumbore intersector          */
/* (kernels/geometry/triangle_intersector_moeller.h, MoellerTrumboreIntersector):
 *   e1 = v0-v1, e2 = v2-v0, Ng = cross(e1,e2); C = v0-O; R = cross(D,C);
 *   den = dot(Ng,D); U = dot(R,e2)^sgn(den); V = dot(R,e1)^sgn(den);
 *   valid = den!=0 & U>=0 & V>=0 & U+V<=|den|;  T = dot(Ng,C)^sgn(den);
 *   valid &= |den|*tnear < T  (& T <= |den|*tfar);  t=T/|den|, u=U/|den|, v=V/|den|.
 * Called from rtcIntersect/rtcOccluded at EmbreeMeshAdapter.cpp:474,375.  Embree divides through
 * an rcp+Newton step; here a true division.  Nearest hit: smaller t wins, equal t -> lower primID
 * (the build's tie rule, independent of traversal order). */
/* ------------------------------------------------------------------------- */
static inline int orc_tri_test(v3 O, v3 D, v3 v0, v3 v1, v3 v2, float tnear, float *t, float *u, float *v) {
  v3 e1 = sub3(v0, v1), e2 = sub3(v2, v0);
  v3 Ng = cross3(e1, e2);
  v3 C = sub3(v0, O);
  v3 R = cross3(D, C);
  float den = dot3(Ng, D);
  float absDen = fabsf(den);
  float sgn = (den < 0.f) ? -1.f : 1.f; /* xor with the sign bit == multiply by +-1 (exact) */
  float U = dot3(R, e2) * sgn;
  float V = dot3(R, e1) * sgn;
  if (!(den != 0.f && U >= 0.f && V >= 0.f && U + V <= absDen)) return 0;
  float T = dot3(Ng, C) * sgn;
  if (!(absDen * tnear < T)) return 0;
  float tt = T / absDen;
  if (!(tt <= FLT_MAX)) return 0; /* T <= |den|*tfar with tfar = FLT_MAX */
  *t = tt; *u = U / absDen; *v = V / absDen;
  return 1;
}

static inline v3 tri_vert(const orc_mesh *M, int32_t prim, int k) { return ld3(M->verts + 3 * M->tris[3 * prim + k]); }

/* slab test of the oracle's own BVH: conservative (padded boxes, 1+3ulp on tfar a la Ize).  The cull against the best hit keeps a relative
 * slack of 2^-10: the DEFINITION of the closest hit is the arg-min of orc_tri_test over ALL triangles (ties to the lower primID; use_bvh = 0 is
 * that loop), and for a ray that grazes a triangle almost in its plane the test's t is noise of relative size 1e-4 -- it can come out EARLIER than
 * the entry into the triangle's own padded box, and a hit found first in a duplicate of that triangle would then cull the box that holds the
 * lower primID (round 5, fuzz seed 531: 12 vertices, 2,379 triangles, rays through vertices: boxes entered at 4.42113 against a best t of 4.42045). */
static inline int box_test(const orc_node *n, v3 O, v3 inv, float tbest, float *tn_out) {
  /* ... and SIDEWAYS: the hit point the test computes for a ray through a vertex or an edge at distance t lies up to ~2^-12 t off the triangle (round 5,
   * soak seeds 2044 / 3662 / 4948 / 8347: a ray credited to the lower primID of two triangles sharing the vertex passes that triangle's exact box 0.04 units
   * away at t = 270).  The definition is the arg-min over ALL triangles, so the box is widened by 2^-11 of the farthest distance the ray can reach inside
   * it; what lies further off its triangle than that is noise no tree is asked to return (the fuzz test's "garbage"). */
  const float fx = fmaxf(fabsf(n->lo[0] - O.x), fabsf(n->hi[0] - O.x)), fy = fmaxf(fabsf(n->lo[1] - O.y), fabsf(n->hi[1] - O.y)),
              fz = fmaxf(fabsf(n->lo[2] - O.z), fabsf(n->hi[2] - O.z));
  const float pad = 0x1p-11f * sqrtf(fx * fx + fy * fy + fz * fz);
  float t0x = (n->lo[0] - pad - O.x) * inv.x, t1x = (n->hi[0] + pad - O.x) * inv.x;
  float t0y = (n->lo[1] - pad - O.y) * inv.y, t1y = (n->hi[1] + pad - O.y) * inv.y;
  float t0z = (n->lo[2] - pad - O.z) * inv.z, t1z = (n->hi[2] + pad - O.z) * inv.z;
  float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.f));
  float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
  tf *= 1.0000004f;
  *tn_out = tn;
  return tn <= tf && tn <= tbest + fabsf(tbest) * 0x1p-10f;
}
/* reciprocal direction for the slab test only; a zero component becomes +-1e-30 so that no NaN appears */
static inline v3 safe_inv(v3 D) {
  float dx = fabsf(D.x) < 1e-30f ? copysignf(1e-30f, D.x) : D.x;
  float dy = fabsf(D.y) < 1e-30f ? copysignf(1e-30f, D.y) : D.y;
  float dz = fabsf(D.z) < 1e-30f ? copysignf(1e-30f, D.z) : D.z;
  return V3(1.0f / dx, 1.0f / dy, 1.0f / dz);
}

static orc_hit closest_hit(const orc_mesh *M, v3 O, v3 D, float tnear, int use_bvh) {
  orc_hit h = { FLT_MAX, -1, 0.f, 0.f };
  float t, u, v;
  if (!use_bvh) {
    for (size_t p = 0; p < M->nT; p++)
      if (orc_tri_test(O, D, tri_vert(M, (int32_t)p, 0), tri_vert(M, (int32_t)p, 1), tri_vert(M, (int32_t)p, 2), tnear, &t, &u, &v))
        if (t < h.t || (t == h.t && (int32_t)p < h.prim) || h.prim < 0) { h.t = t; h.prim = (int32_t)p; h.u = u; h.v = v; }
    return h;
  }
  if (!M->nT) return h;
  v3 inv = safe_inv(D);
  int32_t stack[256];
  float tn0, tn1;
  int sp = 0;
  if (!box_test(&M->nodes[0], O, inv, h.t, &tn0)) return h;
  stack[sp++] = 0;
  while (sp) {
    const orc_node *n = &M->nodes[stack[--sp]];
    if (n->count) {
      for (int32_t i = 0; i < n->count; i++) {
        int32_t p = M->prim_idx[n->left + i];
        if (orc_tri_test(O, D, tri_vert(M, p, 0), tri_vert(M, p, 1), tri_vert(M, p, 2), tnear, &t, &u, &v))
          if (h.prim < 0 || t < h.t || (t == h.t && p < h.prim)) { h.t = t; h.prim = p; h.u = u; h.v = v; }
      }
    } else { /* near child first; a far child is re-tested against the current best when popped late */
      int h0 = box_test(&M->nodes[n->left], O, inv, h.t, &tn0), h1 = box_test(&M->nodes[n->left + 1], O, inv, h.t, &tn1);
      if (h0 && h1) {
        if (tn1 < tn0) { stack[sp++] = n->left; stack[sp++] = n->left + 1; }
        else { stack[sp++] = n->left + 1; stack[sp++] = n->left; }
      } else if (h0) stack[sp++] = n->left;
      else if (h1) stack[sp++] = n->left + 1;
    }
  }
  return h;
}

static int any_hit(const orc_mesh *M, v3 O, v3 D, float tnear, int use_bvh) {
  float t, u, v;
  if (!use_bvh) {
    for (size_t p = 0; p < M->nT; p++)
      if (orc_tri_test(O, D, tri_vert(M, (int32_t)p, 0), tri_vert(M, (int32_t)p, 1), tri_vert(M, (int32_t)p, 2), tnear, &t, &u, &v)) return 1;
    return 0;
  }
  if (!M->nT) return 0;
  v3 inv = safe_inv(D);
  int32_t stack[256];
  float tn0;
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const orc_node *n = &M->nodes[stack[--sp]];
    if (!box_test(n, O, inv, FLT_MAX, &tn0)) continue;
    if (n->count) {
      for (int32_t i = 0; i < n->count; i++) {
        int32_t p = M->prim_idx[n->left + i];
        if (orc_tri_test(O, D, tri_vert(M, p, 0), tri_vert(M, p, 1), tri_vert(M, p, 2), tnear, &t, &u, &v)) return 1;
      }
    } else {
      stack[sp++] = n->left + 1;
      stack[sp++] = n->left;
    }
  }
  return 0;
}

void orc_intersect(const orc_mesh *M, const float *org, const float *dir, size_t n, float tnear, int use_bvh, orc_hit *out) {
  for (size_t i = 0; i < n; i++) out[i] = closest_hit(M, ld3(org + 3 * i), ld3(dir + 3 * i), tnear, use_bvh);
}
void orc_occluded(const orc_mesh *M, const float *org, const float *dir, size_t n, float tnear, int use_bvh, int32_t *out) {
  for (size_t i = 0; i < n; i++) out[i] = any_hit(M, ld3(org + 3 * i), ld3(dir + 3 * i), tnear, use_bvh);
}

/* ------------------------------------------------------------------------- */
/* Adapter::trace restated (EmbreeMeshAdapter.cpp:436-660)                    */
/* ------------------------------------------------------------------------- */
typedef struct {
  orc_ray *v;
  size_t n, cap;
} rayvec;
static void rv_push(rayvec *q, const orc_ray *r) {
  if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 1024; q->v = realloc(q->v, q->cap * sizeof(orc_ray)); }
  q->v[q->n++] = *r;
}
static void rv_append(rayvec *q, const orc_ray *r, size_t n) {
  if (q->n + n > q->cap) { while (q->n + n > q->cap) q->cap = q->cap ? q->cap * 2 : 1024; q->v = realloc(q->v, q->cap * sizeof(orc_ray)); }
  memcpy(q->v + q->n, r, n * sizeof(orc_ray));
  q->n += n;
}

/* CosWeightedRandomHemisphereDirection2 (EmbreeMeshAdapter.cpp:289-318) */
static v3 cos_weighted_dir(v3 n, uint32_t *seed) {
  float Xi1 = fastrand01(seed);
  float Xi2 = fastrand01(seed);
  /* acos / sinf / cosf: the written-out definitions of include/gvt_math.h (libm's differ between machines in the last bit) */
  float theta = (float)gvt_acos(__builtin_sqrt(1.0 - (double)Xi1));
  float phi = (float)(2.0 * 3.1415926535897932384626433832795 * (double)Xi2);
  float xs = gvt_sinf(theta) * gvt_cosf(phi);
  float ys = gvt_cosf(theta);
  float zs = gvt_sinf(theta) * gvt_sinf(phi);
  v3 y = n, h = y;
  if (fabsf(h.x) <= fabsf(h.y) && fabsf(h.x) <= fabsf(h.z)) h.x = 1.0f;
  else if (fabsf(h.y) <= fabsf(h.x) && fabsf(h.y) <= fabsf(h.z)) h.y = 1.0f;
  else h.z = 1.0f;
  v3 x = cross3(h, y);
  v3 z = cross3(x, y);
  v3 d = add3(add3(scl3(x, xs), scl3(y, ys)), scl3(z, zs));
  return norm3(d);
}

void orc_cos_weighted_dir(const float n[3], uint32_t *seed, float out[3]) {
  v3 d = cos_weighted_dir(ld3(n), seed);
  out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void orc_math_probe(int kind, const float *in, size_t n, float *out) {
  for (size_t i = 0; i < n; i++) {
    const float x = in[i];
    switch (kind) {
    case 0: out[i] = gvt_sinf(x); break;
    case 1: out[i] = gvt_cosf(x); break;
    case 2: out[i] = (float)gvt_acos(__builtin_sqrt(1.0 - (double)x)); break;
    case 16: out[i] = sinf(x); break;
    case 17: out[i] = cosf(x); break;
    case 18: out[i] = (float)acos(sqrt(1.0 - (double)x)); break;
    default: out[i] = 0.f;
    }
  }
}

typedef struct {
  const orc_mesh *M;
  orc_ray *rays;
  size_t begin, end;
  const float *m, *minv, *normi;
  const orc_light *lights;
  size_t nLights;
  int normal_mode;
  uint32_t seed;
  int carried_rng; /* 1: a ray's stream is the word it carries; 0: keyed on (seed, index in rayList) */
  rayvec out;
  uint64_t n_closest, n_any;
} trace_job;

static void trace_range(trace_job *J) {
  const orc_mesh *M = J->M;
  orc_ray shadow[64];
  for (size_t idx = J->begin; idx < J->end; idx++) {
    orc_ray *r = &J->rays[idx];
    uint32_t g_seed = (J->carried_rng && ray_word(r) != 0u) ? ray_word(r) : ray_stream_seed(J->seed, idx);
    int alive = 1;
    while (alive) {
      /* prepGVT_EMBREE_PACKET_TYPE :255-287: tnear = RAY_EPSILON, tfar = FLT_MAX, ray t_min/t_max ignored.
       * Instance transform (:633-641) restated as the object-space ray of EmbreeStreamMeshAdapter.cpp:308-309. */
      v3 O = xfm_point(J->minv, ld3(r->origin));
      v3 D = xfm_vector(J->minv, ld3(r->direction));
      orc_hit h = closest_hit(M, O, D, RAY_EPSILON, 1);
      J->n_closest++;
      size_t nShadow = 0;
      if (h.prim < 0) { /* :605-609 */
        rv_push(&J->out, r);
        break;
      }
      if (r->type == 1) break; /* SHADOW ray hit something: dropped :486-488 */
      float t = h.t;
      r->t = t; /* :491 */
      v3 v0 = tri_vert(M, h.prim, 0), v1 = tri_vert(M, h.prim, 1), v2 = tri_vert(M, h.prim, 2);
      /* -Ng of Embree == cross(v1-v0, v2-v0) (cf. OptixMeshAdapter.cu:280-287) */
      v3 negNg = cross3(sub3(v1, v0), sub3(v2, v0));
      v3 normalflat = norm3(mat3_mul(J->normi, negNg)); /* :504 */
      v3 N;
      if (J->normal_mode == 1) { /* smooth, :505-518 */
        v3 a = ld3(M->normals + 3 * M->tris[3 * h.prim + 1]);
        v3 b = ld3(M->normals + 3 * M->tris[3 * h.prim + 2]);
        v3 c = ld3(M->normals + 3 * M->tris[3 * h.prim + 0]);
        v3 mn = add3(add3(scl3(a, h.u), scl3(b, h.v)), scl3(c, 1.0f - h.u - h.v));
        N = norm3(mat3_mul(J->normi, mn));
      } else {
        N = normalflat; /* FLAT_SHADING :520-522 */
      }
      if (dot3(neg3(ld3(r->direction)), normalflat) <= 0.f) N = neg3(N); /* :527-529 */

      orc_material tmp;
      const orc_material *mat;
      if (M->vcolors) { /* :535-561 */
        v3 c0 = ld3(M->vcolors + 3 * M->tris[3 * h.prim]), c1 = ld3(M->vcolors + 3 * M->tris[3 * h.prim + 1]),
           c2 = ld3(M->vcolors + 3 * M->tris[3 * h.prim + 2]);
        v3 ci = add3(add3(scl3(c0, 1.f - h.u - h.v), scl3(c1, h.u)), scl3(c2, h.v));
        default_material(&tmp);
        tmp.type = 0;
        st3(tmp.kd, ci);
        mat = &tmp;
      } else if (M->face_mat && M->face_mat[h.prim] >= 0 && (size_t)M->face_mat[h.prim] < M->nMat) {
        mat = &M->materials[M->face_mat[h.prim]]; /* :563-564 */
      } else {
        mat = &M->mesh_mat; /* :566 */
      }
      if (r->type == 2) { /* SECONDARY :572-575 */
        t = (t > 1) ? 1.f / t : t;
        r->w = r->w * t;
      }
      /* generateShadowRays :320-358 */
      for (size_t li = 0; li < J->nLights && nShadow < 64; li++) {
        const orc_light *L = &J->lights[li];
        v3 lightPos = (L->type == ORC_LIGHT_AREA) ? area_light_position(L, &g_seed) : ld3(L->position);
        float Nf[3], lp[3], c[3];
        st3(Nf, N); st3(lp, lightPos);
        if (!orc_shade(mat, r, Nf, L, lp, c)) continue;
        const float multiplier = 1.0f - RAY_EPSILON * 16;
        const float t_shadow = multiplier * r->t;
        v3 origin = add3(ld3(r->origin), scl3(ld3(r->direction), t_shadow));
        v3 dir = sub3(lightPos, origin);
        orc_ray *s = &shadow[nShadow++];
        memset(s, 0, sizeof *s);
        st3(s->origin, origin);
        st3(s->direction, norm3(dir)); /* Ray ctor normalizes, Ray.h:109 */
        s->t_min = RAY_EPSILON;
        s->w = r->w;
        s->type = 1;
        s->depth = r->depth; /* the ctor leaves depth unset (Ray.h:106-116); restated as the parent's depth */
        s->t = r->t;
        s->id = r->id;
        s->t_max = 3.0f; /* dir.length() == glm component count, :347,355 */
        memcpy(s->color, c, 12);
      }
      int ndepth = r->depth - 1; /* :584-602 */
      float p = 1.f - fastrand01(&g_seed);
      if (ndepth > 0 && r->w > p) {
        r->type = 2;
        const float multiplier = 1.0f - 16.0f * FLT_EPSILON;
        const float t_secondary = multiplier * r->t;
        st3(r->origin, add3(ld3(r->origin), scl3(ld3(r->direction), t_secondary)));
        v3 nd = cos_weighted_dir(N, &g_seed);
        st3(r->direction, nd);
        r->w = r->w * dot3(nd, N);
        r->depth = ndepth;
        km_clear(r); /* a new straight segment: nothing is known about it */
      } else {
        alive = 0;
      }
      set_ray_word(r, g_seed); /* the stream goes on with the ray (rayList is updated in place) */
      /* traceShadowRays :364-385: any-hit in (RAY_EPSILON, FLT_MAX) against the same instance */
      for (size_t s = 0; s < nShadow; s++) {
        v3 so = xfm_point(J->minv, ld3(shadow[s].origin));
        v3 sd = xfm_vector(J->minv, ld3(shadow[s].direction));
        J->n_any++;
        if (!any_hit(M, so, sd, RAY_EPSILON, 1)) rv_push(&J->out, &shadow[s]);
      }
    }
  }
}

/* Persistent worker pool.  Threads are created once and each is pinned to one CPU of the process's affinity mask: threads created
 * per call were observed to share ONE core for the first ~second on this class of machine (the kernel's load balancer had not
 * spread them yet), which made a first multi-threaded call run at single-thread speed. */
typedef struct {
  trace_job *jobs;
  size_t nJobs;
  size_t next;     /* next job to hand out */
  size_t done;     /* jobs finished */
  int limit;       /* workers allowed to take part in the current batch */
  uint64_t batch;  /* generation counter */
} trace_batch;
static struct {
  pthread_mutex_t mu;
  pthread_cond_t work, idle;
  pthread_t th[1024];
  int n;
  trace_batch b;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { 0 }, 0, { 0 } };

static void *pool_worker(void *arg) {
  const int me = (int)(intptr_t)arg;
  pthread_mutex_lock(&g_pool.mu);
  for (;;) {
    while (!(me < g_pool.b.limit && g_pool.b.next < g_pool.b.nJobs)) pthread_cond_wait(&g_pool.work, &g_pool.mu);
    const size_t j = g_pool.b.next++;
    trace_job *J = &g_pool.b.jobs[j];
    pthread_mutex_unlock(&g_pool.mu);
    trace_range(J);
    pthread_mutex_lock(&g_pool.mu);
    if (++g_pool.b.done == g_pool.b.nJobs) pthread_cond_signal(&g_pool.idle);
  }
  return NULL;
}
static void pool_run(trace_job *jobs, size_t nJobs, int nthreads) {
  if (nthreads > 1024) nthreads = 1024;
  pthread_mutex_lock(&g_pool.mu);
  if (g_pool.n < nthreads) {
    cpu_set_t allowed;
    int ncpu = 0, cpus[1024];
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
      for (int c = 0; c < CPU_SETSIZE && ncpu < 1024; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    while (g_pool.n < nthreads) {
      const int k = g_pool.n;
      if (pthread_create(&g_pool.th[k], NULL, pool_worker, (void *)(intptr_t)k) != 0) break;
      if (ncpu > 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[k % ncpu], &one); pthread_setaffinity_np(g_pool.th[k], sizeof one, &one); }
      pthread_detach(g_pool.th[k]);
      g_pool.n++;
    }
  }
  g_pool.b.jobs = jobs; g_pool.b.nJobs = nJobs; g_pool.b.next = 0; g_pool.b.done = 0; g_pool.b.limit = nthreads < g_pool.n ? nthreads : g_pool.n;
  g_pool.b.batch++;
  pthread_cond_broadcast(&g_pool.work);
  while (g_pool.b.done < nJobs) pthread_cond_wait(&g_pool.idle, &g_pool.mu);
  g_pool.b.nJobs = 0; g_pool.b.limit = 0;
  pthread_mutex_unlock(&g_pool.mu);
}

static uint64_t g_last_closest, g_last_any;
void orc_trace_counts(uint64_t *c, uint64_t *a) { *c = g_last_closest; *a = g_last_any; }

static void trace_to_vec(const orc_mesh *M, orc_ray *rays, size_t begin, size_t end, rayvec *out, const float *m,
                         const float *minv, const float *normi, const orc_light *lights, size_t nLights, int normal_mode,
                         uint32_t seed, int nthreads, int carried_rng) {
  if (end == 0 || end < begin) end = begin; /* caller resolves end==0 -> size */
  size_t n = end - begin;
  size_t chunk = 4096; /* work grain of EmbreeMeshAdapter.cpp:648 */
  size_t nJobs = (n + chunk - 1) / chunk;
  if (nthreads < 1) nthreads = 1;
  trace_job *jobs = calloc(nJobs ? nJobs : 1, sizeof *jobs);
  for (size_t j = 0; j < nJobs; j++) {
    trace_job *J = &jobs[j];
    J->M = M; J->rays = rays; J->begin = begin + j * chunk;
    J->end = (J->begin + chunk < end) ? J->begin + chunk : end;
    J->m = m; J->minv = minv; J->normi = normi; J->lights = lights; J->nLights = nLights;
    J->normal_mode = normal_mode; J->seed = seed; J->carried_rng = carried_rng;
  }
  if (nthreads == 1 || nJobs <= 1) {
    for (size_t j = 0; j < nJobs; j++) trace_range(&jobs[j]);
  } else {
    pool_run(jobs, nJobs, nthreads);
  }
  g_last_closest = g_last_any = 0;
  for (size_t j = 0; j < nJobs; j++) { /* chunk order: deterministic output order */
    rv_append(out, jobs[j].out.v, jobs[j].out.n);
    free(jobs[j].out.v);
    g_last_closest += jobs[j].n_closest;
    g_last_any += jobs[j].n_any;
  }
  free(jobs);
}

int orc_trace(const orc_mesh *M, orc_ray *rays, size_t begin, size_t end, orc_ray *rays_out, size_t cap, size_t *n_out,
              const float m[16], const float minv[16], const float normi[9], const orc_light *lights, size_t nLights,
              int normal_mode, uint32_t seed, int nthreads) {
  rayvec out = { 0 };
  trace_to_vec(M, rays, begin, end, &out, m, minv, normi, lights, nLights, normal_mode, seed, nthreads, 0);
  *n_out = out.n;
  int rc = 0;
  if (out.n > cap) rc = -1;
  else if (out.n) memcpy(rays_out, out.v, out.n * sizeof(orc_ray));
  free(out.v);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Camera (data/scene/gvtCamera.cpp:89-171 RIGHT_HAND_CAMERA, 233-312)        */
/* ------------------------------------------------------------------------- */
void orc_camera_generate(const float eye[3], const float focus[3], const float up[3], float fov, int W, int H,
                         int samples, int depth, float jitterWindowSizeF, orc_ray *rays) {
  v3 e = ld3(eye), f = ld3(focus), upv = ld3(up);
  v3 w = norm3(sub3(f, e));
  v3 v = norm3(upv);
  v3 u;
  u.x = w.y * v.z - w.z * v.y; u.y = w.z * v.x - w.x * v.z; u.z = w.x * v.y - w.y * v.x;
  u = norm3(u);
  v3 up2;
  up2.x = u.y * w.z - w.y * u.z; up2.y = u.z * w.x - w.z * u.x; up2.z = u.x * w.y - w.x * u.y;
  v = norm3(up2);
  int jitterWindowSize = (int)jitterWindowSizeF; /* setJitterWindowSize(int) truncates, gvtCamera.cpp:200 */
  float aspectRatio = (float)W / (float)H;
  const float vert = tanf((float)(fov * 0.5));
  const float horz = tanf((float)(fov * 0.5)) * aspectRatio;
  const float divider = (float)samples;
  const float offset = (float)((1.0 / divider) * jitterWindowSize);
  const float wmult = 2.f / (float)(W - 1);
  const float hmult = 2.f / (float)(H - 1);
  const float half_sample = samples * 0.5f;
  const size_t samples2 = (size_t)samples * samples;
  const float contri = 1.f / (samples * samples);
  for (int j = 0; j < H; j++) {
    int idx = j * W;
    for (int i = 0; i < W; i++) {
      const float x0 = (float)((float)i * wmult - 1.0), y0 = (float)((float)j * hmult - 1.0);
      for (int k = 0; k < samples; k++)
        for (int ww = 0; ww < samples; ww++) {
          size_t ridx = (size_t)idx * samples2 + (size_t)k * samples + ww;
          float x = x0 + (ww - half_sample) * offset;
          x *= horz;
          float y = y0 + (k - half_sample) * offset;
          y *= vert;
          v3 d;
          d.x = u.x * x + v.x * y + w.x;
          d.y = u.y * x + v.y * y + w.y;
          d.z = u.z * x + v.z * y + w.z;
          orc_ray *r = &rays[ridx];
          memset(r, 0, sizeof *r);
          r->id = idx;
          r->t_min = RAY_EPSILON;
          st3(r->origin, e);
          st3(r->direction, norm3(d));
          r->t_max = FLT_MAX;
          r->t = FLT_MAX;
          r->w = contri;
          r->type = 0;
          r->depth = depth;
          set_ray_word(r, camera_stream_word(ridx));
        }
      idx++;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Top-level BVH over instances (accel/BVH.cpp:77-216)                        */
/* ------------------------------------------------------------------------- */
typedef struct {
  const float *lo, *hi;
  int32_t *set; /* instanceSet permutation */
  int32_t *sorted;
  size_t nSorted;
} top_builder;

static inline float top_centroid(const top_builder *B, int32_t inst, int ax) { /* BBox.cpp:128 */
  return 0.5f * B->lo[3 * inst + ax] + 0.5f * B->hi[3 * inst + ax];
}
static void box_merge(float lo[3], float hi[3], const float *olo, const float *ohi) {
  for (int k = 0; k < 3; k++) { lo[k] = fmin_ref(olo[k], lo[k]); hi[k] = fmax_ref(ohi[k], hi[k]); }
}
static float box_area(const float lo[3], const float hi[3]) { /* BBox.cpp:130-133 */
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return (2.f * (dx * dy + dy * dz + dz * dx));
}
static float top_split_point(top_builder *B, int ax, int start, int end) { /* BVH.cpp:173-216 */
  float minCost = FLT_MAX, splitPoint = 0.f;
  for (int i = start; i < end; ++i) {
    for (int e = 0; e < 2; ++e) {
      float edge = (e == 0) ? B->lo[3 * B->set[i] + ax] : B->hi[3 * B->set[i] + ax];
      float llo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lhi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
      float rlo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, rhi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
      int leftCount = 0;
      for (int j = start; j < end; ++j) {
        int32_t q = B->set[j];
        if (top_centroid(B, q, ax) < edge) { ++leftCount; box_merge(llo, lhi, B->lo + 3 * q, B->hi + 3 * q); }
        else box_merge(rlo, rhi, B->lo + 3 * q, B->hi + 3 * q);
      }
      int rightCount = end - start - leftCount;
      float cost = (float)(0.5 + (box_area(llo, lhi) * leftCount) + (box_area(rlo, rhi) * rightCount));
      if (cost < minCost) { minCost = cost; splitPoint = edge; }
    }
  }
  return splitPoint;
}
static void top_build(top_builder *B, int start, int end) { /* BVH.cpp:77-171, LEAF_SIZE 1 */
  float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
  for (int i = start; i < end; ++i) box_merge(lo, hi, B->lo + 3 * B->set[i], B->hi + 3 * B->set[i]);
  int count = end - start;
  if (count <= 1) { for (int i = start; i < end; ++i) B->sorted[B->nSorted++] = B->set[i]; return; }
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  int ax = (dx > dy && dx > dz) ? 0 : (dy > dz) ? 1 : 2; /* wideRangingBoxDir BBox.cpp:117-126 */
  float sp = top_split_point(B, ax, start, end);
  /* std::partition, libstdc++ bidirectional form (stl_algo.h __partition) */
  int first = start, last = end;
  for (;;) {
    for (;;) { if (first == last) goto done; else if (top_centroid(B, B->set[first], ax) < sp) ++first; else break; }
    --last;
    for (;;) { if (first == last) goto done; else if (!(top_centroid(B, B->set[last], ax) < sp)) --last; else break; }
    int32_t t = B->set[first]; B->set[first] = B->set[last]; B->set[last] = t;
    ++first;
  }
done:;
  int splitIdx = first;
  if (splitIdx == start || splitIdx == end) { for (int i = start; i < end; ++i) B->sorted[B->nSorted++] = B->set[i]; return; }
  top_build(B, start, splitIdx);
  top_build(B, splitIdx, end);
}
void orc_toplevel_order(const float *lo, const float *hi, size_t n, int32_t *order) {
  top_builder B = { lo, hi, malloc(sizeof(int32_t) * (n + 1)), order, 0 };
  for (size_t i = 0; i < n; i++) B.set[i] = (int32_t)i;
  top_build(&B, 0, (int)n);
  free(B.set);
}

/* BVH::intersect leaf step (BVH.h:103-117) with RayPacketIntersection::intersect(update=true)
 * (RayPacket.h:111-211) restated per ray: inner-node tests only prune, and an instance box that
 * passes implies that every enclosing node box passes, so visiting the leaves in DFS order is
 * equivalent. */
static inline int top_one(const orc_ray *r, const float *lo, const float *hi, const int32_t *order, size_t nInst,
                          int from, float *t_out) {
  float ox = r->origin[0], oy = r->origin[1], oz = r->origin[2];
  float dx = 1.f / r->direction[0], dy = 1.f / r->direction[1], dz = 1.f / r->direction[2];
  float t = r->t_max;
  float ret_t = FLT_MAX;
  int next = -1;
  for (size_t k = 0; k < nInst; k++) {
    int32_t q = order[k];
    if (from == q) continue;
    const float *bl = lo + 3 * q, *bu = hi + 3 * q;
    float lx = (bl[0] - ox) * dx, ly = (bl[1] - oy) * dy, lz = (bl[2] - oz) * dz;
    float ux = (bu[0] - ox) * dx, uy = (bu[1] - oy) * dy, uz = (bu[2] - oz) * dz;
    float minx = fmin_ref(lx, ux), maxx = fmax_ref(lx, ux);
    float miny = fmin_ref(ly, uy), maxy = fmax_ref(ly, uy);
    float minz = fmin_ref(lz, uz), maxz = fmax_ref(lz, uz);
    float tnear = fmax_ref(fmax_ref(minx, miny), minz);
    float tfar = fmin_ref(fmin_ref(maxx, maxy), maxz);
    int hit = (tfar > tnear && tnear > RAY_EPSILON && t > tnear);
    if (hit) {
      t = tnear;
      if (ret_t > t) { next = q; ret_t = t; }
    }
  }
  *t_out = ret_t;
  return next;
}
void orc_toplevel_intersect(const float *lo, const float *hi, const int32_t *order, size_t nInst, const orc_ray *rays,
                            size_t n, int from, int32_t *next_out, float *t_out) {
  for (size_t i = 0; i < n; i++) next_out[i] = top_one(&rays[i], lo, hi, order, nInst, from, &t_out[i]);
}

/* ------------------------------------------------------------------------- */
/* Framebuffer (composite/IceTComposite.cpp:79-157)                           */
/* ------------------------------------------------------------------------- */
static inline void fb_local_add(float *fb, size_t idx, v3 color, float alpha) { /* :111-117 */
  float c[3] = { color.x, color.y, color.z };
  for (int i = 0; i < 3; i++) {
    fb[idx * 4 + i] += c[i];
    if (fb[idx * 4 + i] > 1.f) fb[idx * 4 + i] = 1.f;
  }
  fb[idx * 4 + 3] += alpha;
}
void orc_fb_to_ppm_bytes(const float *fb, int W, int H, unsigned char *out) { /* :119-157 */
  size_t o = 0;
  for (int j = H - 1; j >= 0; j--)
    for (int i = 0; i < W; ++i) {
      size_t index = 4 * ((size_t)j * W + i);
      out[o++] = (unsigned char)(fb[index + 0] * 255);
      out[o++] = (unsigned char)(fb[index + 1] * 255);
      out[o++] = (unsigned char)(fb[index + 2] * 255);
    }
}

/* ------------------------------------------------------------------------- */
/* Schedulers                                                                 */
/* ------------------------------------------------------------------------- */
/* shuffleRays' decision for ONE ray leaving instance `from` (TracerBase.h:392-400): the instance it goes on in (its origin advanced,
 * :393) or -1.  With the known-miss shortcut: `from` joins the ray's list, and choices that are on the list are walked through. */
#define KM_MAX_HOPS 64
static int shuffle_one(const float *lo, const float *hi, const int32_t *order, size_t nInst, orc_ray *r, int from) {
  float t;
  int next = top_one(r, lo, hi, order, nInst, from, &t);
  if (!g_skip_known) {
    if (next != -1) st3(r->origin, add3(ld3(r->origin), scl3(ld3(r->direction), t * 0.95f)));
    return next;
  }
  uint16_t e[6];
  km_get(r, e);
  km_add(e, from);
  for (int hop = 0; next != -1; hop++) {
    st3(r->origin, add3(ld3(r->origin), scl3(ld3(r->direction), t * 0.95f)));
    if (!km_has(e, next) || hop >= KM_MAX_HOPS) break;
    from = next; /* crossed without a hit before: as if traced there again and forwarded */
    next = top_one(r, lo, hi, order, nInst, from, &t);
  }
  km_put(r, e);
  return next;
}
/* the same for n rays (in place), for the checker backend of the Python scheduler harness */
void orc_shuffle_step(const float *lo, const float *hi, const int32_t *order, size_t nInst, orc_ray *rays, size_t n, int from, int32_t *next_out) {
  for (size_t i = 0; i < n; i++) next_out[i] = shuffle_one(lo, hi, order, nInst, &rays[i], from);
}
/* shuffleRays non-volume branch (TracerBase.h:325-343, 392-414) */
static void shuffle_rays(const orc_scene *S, const int32_t *order, rayvec *rays, int domID, rayvec *queues, float *fb) {
  for (size_t i = 0; i < rays->n; i++) {
    orc_ray *r = &rays->v[i];
    int next = shuffle_one(S->inst_lo, S->inst_hi, order, S->nInst, r, domID);
    if (next != -1) {
      rv_push(&queues[next], r);
    } else if (r->type == 1 && len3(ld3(r->color)) > 0) {
      fb_local_add(fb, (size_t)r->id, scl3(ld3(r->color), r->w), 1.f);
    }
  }
  rays->n = 0;
}

void orc_render_image(const orc_scene *S, orc_ray *cam, size_t nRays, int W, int H, float *fb, orc_frame_stats *st) {
  memset(fb, 0, sizeof(float) * 4 * (size_t)W * H); /* clearBuffer -> IceTComposite::reset :79-82 */
  int32_t *order = malloc(sizeof(int32_t) * (S->nInst + 1));
  orc_toplevel_order(S->inst_lo, S->inst_hi, S->nInst, order);
  rayvec *queues = calloc(S->nInst + 1, sizeof(rayvec));
  rayvec rays = { cam, nRays, nRays }, moved = { 0 };
  orc_frame_stats s = { 0 };
  shuffle_rays(S, order, &rays, -1, queues, fb); /* FilterRaysLocally ImageTracer.h:111-125 */
  for (;;) { /* ImageTracer.h:159-259 */
    int target = -1;
    size_t cnt = 0;
    for (size_t q = 0; q < S->nInst; q++)
      if (queues[q].n > cnt) { cnt = queues[q].n; target = (int)q; }
    if (target < 0) break;
    trace_to_vec(S->meshes[target], queues[target].v, 0, queues[target].n, &moved, S->m + 16 * target,
                 S->minv + 16 * target, S->normi + 9 * target, S->lights, S->nLights, S->normal_mode,
                 (uint32_t)s.adapter_calls, S->nthreads, 1);
    s.rays_closest += g_last_closest; s.rays_any += g_last_any; s.adapter_calls++;
    queues[target].n = 0;
    shuffle_rays(S, order, &moved, target, queues, fb);
  }
  for (size_t q = 0; q < S->nInst; q++) free(queues[q].v);
  free(queues); free(moved.v); free(order);
  if (st) *st = s;
}

/* Tracer<DomainScheduler> (DomainTracer.h:115-496) simulated over P virtual ranks in one process.
 * Composite: per-rank float framebuffers summed then clamped to 1 (equals the 1-rank image
 * whenever at most one rank writes a pixel; IceT is not in the tree). */
void orc_render_domain(const orc_scene *S, const int32_t *owner, int P, const orc_ray *cam, size_t nRays, int W, int H,
                       float *fb, orc_frame_stats *st) {
  size_t npx = (size_t)W * H;
  int32_t *order = malloc(sizeof(int32_t) * (S->nInst + 1));
  orc_toplevel_order(S->inst_lo, S->inst_hi, S->nInst, order);
  float **rfb = malloc(sizeof(float *) * (size_t)P);
  rayvec **rq = malloc(sizeof(rayvec *) * (size_t)P);
  orc_frame_stats s = { 0 };
  for (int p = 0; p < P; p++) {
    rfb[p] = calloc(npx * 4, sizeof(float));
    rq[p] = calloc(S->nInst + 1, sizeof(rayvec));
    /* shuffleDropRays :148-183: every rank tests all camera rays, keeps the ones whose first domain is local */
    for (size_t i = 0; i < nRays; i++) {
      orc_ray r = cam[i];
      float t;
      int next = top_one(&r, S->inst_lo, S->inst_hi, order, S->nInst, -1, &t);
      if (next != -1) {
        st3(r.origin, add3(ld3(r.origin), scl3(ld3(r.direction), t * 0.95f)));
        if (owner[next] == p) rv_push(&rq[p][next], &r);
      }
    }
  }
  rayvec moved = { 0 };
  for (;;) {
    for (int p = 0; p < P; p++) { /* each rank: trace until its local queues are dry :228-326 */
      for (;;) {
        int target = -1;
        size_t cnt = 0;
        for (size_t q = 0; q < S->nInst; q++)
          if (owner[q] == p && rq[p][q].n > cnt) { cnt = rq[p][q].n; target = (int)q; }
        if (target < 0) break;
        trace_to_vec(S->meshes[target], rq[p][target].v, 0, rq[p][target].n, &moved, S->m + 16 * target,
                     S->minv + 16 * target, S->normi + 9 * target, S->lights, S->nLights, S->normal_mode,
                     (uint32_t)s.adapter_calls, S->nthreads, 1);
        s.rays_closest += g_last_closest; s.rays_any += g_last_any; s.adapter_calls++;
        rq[p][target].n = 0;
        shuffle_rays(S, order, &moved, target, rq[p], rfb[p]);
      }
    }
    /* SendRays :370-496: every non-local queue moves to its owner */
    for (int p = 0; p < P; p++)
      for (size_t q = 0; q < S->nInst; q++)
        if (owner[q] != p && rq[p][q].n) {
          rv_append(&rq[owner[q]][q], rq[p][q].v, rq[p][q].n);
          s.rays_sent += rq[p][q].n;
          rq[p][q].n = 0;
        }
    s.rounds++;
    size_t not_done = 0; /* :337-349 */
    for (int p = 0; p < P; p++)
      for (size_t q = 0; q < S->nInst; q++) not_done += rq[p][q].n;
    if (!not_done) break;
  }
  for (size_t i = 0; i < npx * 4; i++) {
    float a = 0.f;
    for (int p = 0; p < P; p++) a += rfb[p][i];
    fb[i] = ((i & 3) != 3 && a > 1.f) ? 1.f : a;
  }
  for (int p = 0; p < P; p++) {
    for (size_t q = 0; q < S->nInst; q++) free(rq[p][q].v);
    free(rq[p]); free(rfb[p]);
  }
  free(rq); free(rfb); free(moved.v); free(order);
  if (st) *st = s;
}
