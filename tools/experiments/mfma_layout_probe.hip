// Which lane / register holds D_blk[m][n] of v_mfma_f32_16x16x1_4b_f32, and which lanes feed A_blk[m] / B_blk[n]?  The layout
// newton_direction_wave (csrc/mw_phys.hpp) assumes: A_blk[m] and B_blk[n] from lane 16 blk + m / n; D_blk[m][n] in lane 16 (m / 4) + n,
// register 4 blk + m % 4.  Prints the number of entries that contradict it (0 = as assumed).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/experiments/_build/mfma_layout_probe tools/experiments/mfma_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x, blk = lane >> 4, i = lane & 15;
    const float a = 1000.0f * (blk + 1) + (i + 1), b = 0.001f * (blk + 1) + 0.01f * (i + 1);
    f16v acc;
    for (int k = 0; k < 16; k++) acc[k] = 0;
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0);
    for (int k = 0; k < 16; k++) out[lane * 16 + k] = acc[k];
    // DPP row_newbcast:5 -- lane 5 of every 16-lane row to the whole row (blk_bcast of newton_direction_wave)
    const float v = 7.0f * lane + 1.0f;
    out[64 * 16 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x155, 0xf, 0xf, false));
    // DPP row_ror:4 -- lane i reads lane (i + 4) % 16 of its row or (i - 4): either way a rotation inside the row (blk_sum only needs that)
    out[64 * 17 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
}
int main() {
    float* d; hipMalloc(&d, 64 * 18 * 4);
    probe<<<1, 64>>>(d);
    float h[64 * 18]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 4; blk++) for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
        const float want = (1000.0f * (blk + 1) + (m + 1)) * (0.001f * (blk + 1) + 0.01f * (n + 1));
        const float got = h[(16 * (m / 4) + n) * 16 + 4 * blk + m % 4];
        if (got != want) { if (bad < 5) printf("blk %d m %d n %d: got %g want %g\n", blk, m, n, got, want); bad++; }
    }
    printf("mfma 16x16x1 4-block layout: %d of 1024 entries contradict the assumed layout\n", bad);
    int badb = 0, badr = 0;
    for (int l = 0; l < 64; l++) {
        if (h[64 * 16 + l] != 7.0f * ((l & 48) | 5) + 1.0f) badb++;
        const float r = h[64 * 17 + l], a = 7.0f * ((l & 48) | ((l + 4) & 15)) + 1.0f, b = 7.0f * ((l & 48) | ((l - 4) & 15)) + 1.0f;
        if (r != a && r != b) badr++;
    }
    printf("dpp row_newbcast:5: %d of 64 lanes wrong; row_ror:4: %d of 64 lanes not a rotation inside the row\n", badb, badr);
    return bad != 0 || badb != 0 || badr != 0;
}
