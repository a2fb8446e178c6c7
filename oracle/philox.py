"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy --
TEST INFRASTRUCTURE: the checker for the counter-based draws of libpixelnerf_hip.so (pnr_raysrc.h).  The reference
(sxyu/pixel-nerf) has no such generator (it calls torch.rand, src/render/nerf.py:111,135,141,158); the pin is the
algorithm's published known-answer vectors (tests/test_hip_rng.py), then this restatement checks the mapping
(seed, ray id, draw, index) -> value that the kernels implement."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """counter (...,4) uint32, key (...,2) uint32 -> (...,4) uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def _block(seed, ray_ids, blk, draw):
    ray_ids = np.asarray(ray_ids, dtype=np.uint64)
    ctr = np.stack([ray_ids & np.uint64(0xFFFFFFFF), ray_ids >> np.uint64(32), np.full_like(ray_ids, blk), np.full_like(ray_ids, draw)], -1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64), ray_ids.shape + (2,))
    return philox4x32_10(ctr.astype(np.uint32), key.astype(np.uint32))


def uniforms(seed, ray_ids, draw, n):
    """(R, n) float32: value i of uniform draw `draw` (0 = u1, 1 = u2, 2 = u3) for every ray id."""
    out = np.empty((len(ray_ids), n), np.float32)
    for b in range((n + 3) // 4):
        v = _block(seed, ray_ids, b, draw)
        w = min(4, n - 4 * b)
        out[:, 4 * b:4 * b + w] = ((v[:, :w] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))
    return out


def normals(seed, ray_ids, n):
    """(R, n) float64 Box-Muller values of the depth-sample draw (draw id 3); the kernels evaluate the same formula in fp32."""
    out = np.empty((len(ray_ids), n), np.float64)
    for b in range((n + 1) // 2):
        v = _block(seed, ray_ids, b, 3)
        for half in range(min(2, n - 2 * b)):
            a, c = v[:, 2 * half], v[:, 2 * half + 1]
            ua = ((a >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
            ub = (c >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
            out[:, 2 * b + half] = np.sqrt(-2.0 * np.log(ua)) * np.cos(2.0 * np.pi * ub)
    return out
