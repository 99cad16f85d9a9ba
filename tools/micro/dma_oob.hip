// Micro-test: what does an out-of-range `buffer_load_dwordx4 ... lds` lane write to LDS (zeros or nothing)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    // lanes with (lane % 3 == 0) are out of range
    int off = (lane % 3 == 0) ? (int)0x80000000 : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    // second piece via global_load_lds
    const float* g = src + 256 + lane * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    float h[512], *d, *o, r[512];
    for (int i = 0; i < 512; ++i) h[i] = 1.f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int zeros = 0, kept = 0, good = 0, bad = 0, g2 = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        float v = r[l * 4 + e];
        if (l % 3 == 0) { if (v == 0.f) zeros++; else if (v == -7.f) kept++; else bad++; }
        else { if (v == h[l * 4 + e]) good++; else bad++; }
        if (r[256 + l * 4 + e] == h[256 + l * 4 + e]) g2++;
    }
    printf("OOB lanes: zeros=%d kept_sentinel=%d | in-range good=%d bad=%d | global_load_lds good=%d/256\n", zeros, kept, good, bad, g2);
    return 0;
}
