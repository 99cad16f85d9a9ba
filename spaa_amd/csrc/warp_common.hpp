// warp_common.hpp — sampling-grid helpers shared by warp.hip (attack path) and train_ops.hip (training path).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// torch.linspace(start, end, steps)[i] as computed by ATen's CPU kernel (symmetric halves)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

struct Bilinear {
    int x0, y0;          // north-west tap
    float nw, ne, sw, se;
};

// ATen CPU bilinear grid_sample (GridSamplerKernel.cpp): unnormalise with align_corners=True, floor, weights
// w = x - floor(x), e = 1 - w, n = y - floor(y), s = 1 - n.
__device__ __forceinline__ Bilinear bilinear_setup(float gx, float gy, int W, int H) {
    const float x = (gx + 1.f) * (0.5f * (float)(W - 1));
    const float y = (gy + 1.f) * (0.5f * (float)(H - 1));
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.f - w, n = y - yn, s = 1.f - n;
    Bilinear b;
    b.x0 = (int)xw;
    b.y0 = (int)yn;
    b.nw = e * s;
    b.ne = w * s;
    b.sw = e * n;
    b.se = w * n;
    return b;
}

}  // namespace
