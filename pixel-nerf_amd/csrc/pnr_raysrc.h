// pnr_raysrc.h -- where the renderer kernels get their rays and their random draws from (gfx950).
//
//   RaySrc   : either an explicit (R,8) ray array (the reference's interface, nerf.py:255-262) or a camera description
//              from which every kernel regenerates the ray it needs -- util.gen_rays (src/util/util.py:113-143,238-276)
//              evaluated in place, in the same fp32 operation order as gen_rays_kernel, so both forms give the same bits.
//              No 32 B/ray array is written or read.
//   NoiseSrc : either the explicit pre-drawn tensors (u1,u2,u3,n4 -- parity tests, torch-generator compatibility) or a
//              counter-based generator: Philox4x32-10 keyed by a 64-bit seed, counter = (ray id, value index / 4, draw).
//              A draw depends only on (seed, GLOBAL ray id, draw, index): chunking, sharding across GPUs and the
//              launch geometry do not change the image (the reference's four torch.rand launches, nerf.py:111,135,141,158,
//              and their 1216 B/ray of HBM traffic at 64+128 go away).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnr {

struct RaySrc {
    const float *rays;   // (R,8) [ox,oy,oz,dx,dy,dz,near,far], or NULL: generate from the camera below
    const float *poses;  // (NV,4,4) camera-to-world, rays ordered (view, row, column)
    int W, H;
    float fx, fy, cx, cy, z_near, z_far;
};

struct Ray8 {
    float ox, oy, oz, dx, dy, dz, near, far;
};

#pragma clang fp contract(off)  // the reference's separately rounded operations (and gen_rays_kernel's)
__device__ __forceinline__ Ray8 load_ray(const RaySrc &s, long long r) {
    Ray8 o;
    if (s.rays) {
        const float *p = s.rays + (size_t)r * 8;
        o.ox = p[0]; o.oy = p[1]; o.oz = p[2]; o.dx = p[3]; o.dy = p[4]; o.dz = p[5]; o.near = p[6]; o.far = p[7];
        return o;
    }
    const int px = (int)(r % s.W), py = (int)((r / s.W) % s.H);
    const long long n = r / ((long long)s.W * s.H);
    const float X = ((float)px - s.cx) / s.fx, Y = ((float)py - s.cy) / s.fy;
    float d0 = X, d1 = -Y, d2 = -1.f;
    const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 /= nrm; d1 /= nrm; d2 /= nrm;
    const float *P = s.poses + (size_t)n * 16;
    o.ox = P[3]; o.oy = P[7]; o.oz = P[11];
    o.dx = P[0] * d0 + P[1] * d1 + P[2] * d2;
    o.dy = P[4] * d0 + P[5] * d1 + P[6] * d2;
    o.dz = P[8] * d0 + P[9] * d1 + P[10] * d2;
    o.near = s.z_near; o.far = s.z_far;
    return o;
}
// near / far only (sampling and compositing kernels)
__device__ __forceinline__ void load_bounds(const RaySrc &s, long long r, float &near, float &far) {
    if (s.rays) { near = s.rays[(size_t)r * 8 + 6]; far = s.rays[(size_t)r * 8 + 7]; }
    else { near = s.z_near; far = s.z_far; }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------- counter-based random draws
enum Draw { DRAW_U1 = 0, DRAW_U2 = 1, DRAW_U3 = 2, DRAW_N4 = 3 };

struct NoiseSrc {
    const float *u1, *u2, *u3, *n4;  // explicit tensors (row-major per ray); all NULL: generate
    uint32_t seed_lo, seed_hi;
    long long id_offset;             // global id of ray r: (r / per_obj) * id_stride + r % per_obj + id_offset
    int id_stride, per_obj;          // (id_stride = rays per object of the WHOLE image when this call renders a shard)
};

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11); known-answer tested in
// tests/test_hip_rng.py against the paper's vectors through pnr_philox_raw
__device__ __host__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
        const U4 n = {(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
        c = n;
        k0 += W0; k1 += W1;
    }
    return c;
}

__device__ __forceinline__ long long ray_id(const NoiseSrc &n, long long r) {
    return (r / n.per_obj) * (long long)n.id_stride + r % n.per_obj + n.id_offset;
}
__device__ __forceinline__ uint32_t pick(const U4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// value i of uniform draw `draw` of ray r: 24 random bits -> [0, 1) exactly representable in fp32, never 1
__device__ __forceinline__ float gen_uniform(const NoiseSrc &n, int draw, long long r, int i) {
    const long long id = ray_id(n, r);
    const U4 c = {(uint32_t)id, (uint32_t)((unsigned long long)id >> 32), (uint32_t)(i >> 2), (uint32_t)draw};
    return (float)(pick(philox4x32_10(c, n.seed_lo, n.seed_hi), i & 3) >> 8) * 0x1p-24f;
}
// value i of the normal draw (Box-Muller on one half of a Philox block): N(0,1)
__device__ __forceinline__ float gen_normal(const NoiseSrc &n, long long r, int i) {
    const long long id = ray_id(n, r);
    const U4 c = {(uint32_t)id, (uint32_t)((unsigned long long)id >> 32), (uint32_t)(i >> 1), (uint32_t)DRAW_N4};
    const U4 v = philox4x32_10(c, n.seed_lo, n.seed_hi);
    const uint32_t a = (i & 1) ? v.z : v.x, b = (i & 1) ? v.w : v.y;
    const float ua = (float)((a >> 8) + 1u) * 0x1p-24f;  // (0, 1]
    const float ub = (float)(b >> 8) * 0x1p-24f;         // [0, 1)
    return sqrtf(-2.f * logf(ua)) * cosf(6.28318530717958647692f * ub);
}

__device__ __forceinline__ float noise_u1(const NoiseSrc &n, long long r, int i, int Kc) {
    return n.u1 ? n.u1[(size_t)r * Kc + i] : gen_uniform(n, DRAW_U1, r, i);
}
__device__ __forceinline__ float noise_u2(const NoiseSrc &n, long long r, int j, int Kimp) {
    return n.u2 ? n.u2[(size_t)r * Kimp + j] : gen_uniform(n, DRAW_U2, r, j);
}
__device__ __forceinline__ float noise_u3(const NoiseSrc &n, long long r, int j, int Kimp) {
    return n.u3 ? n.u3[(size_t)r * Kimp + j] : gen_uniform(n, DRAW_U3, r, j);
}
__device__ __forceinline__ float noise_n4(const NoiseSrc &n, long long r, int j, int Kfd) {
    return n.n4 ? n.n4[(size_t)r * Kfd + j] : gen_normal(n, r, j);
}

}  // namespace pnr
