// Probe for the lane-form Poseidon's circulant layer on the matrix pipe (gfx950): one permutation per lane, the 12 x 12 MDS matrix as a
// block-diagonal 32 x 32 i8 weight tile, the state's eight byte planes as the other operand of v_mfma_i32_32x32x32_i8.
//   * checks the operand / result lane maps the scheme relies on with exact integer data (all 64 lanes, random 64-bit words);
//   * times N rounds of (8 MFMA + V independent v_mad_u64_u32) against the same loop without the MFMAs: what an MFMA costs the vector issue.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe tools/experiments/mfma_mds_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const int CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
// plonky2's layer: out[r] = sum_i CIRC[i] * in[(i + r) % 12] + 8 * in[0] for r = 0  ->  M[r][j] = CIRC[(j - r) mod 12] (+ 8 at [0][0])
__host__ __device__ inline int mds(int r, int j) { return CIRC[((j - r) % 12 + 12) % 12] + ((r == 0 && j == 0) ? 8 : 0); }

__device__ inline v4i weights_for_lane(unsigned lane) {
    const unsigned row = lane & 31u, h = lane >> 5;
    const unsigned g = (row & 3u) + 4u * (row >> 3);
    uint8_t w[16];
    for (int j = 0; j < 16; j++) w[j] = 0;
    if (((row >> 2) & 1u) == h && g < 12)
        for (int j = 0; j < 12; j++) w[j] = (uint8_t)mds((int)g, j);
    v4i a;
    for (int d = 0; d < 4; d++) a[d] = (int)(w[4 * d] | (w[4 * d + 1] << 8) | (w[4 * d + 2] << 16) | ((uint32_t)w[4 * d + 3] << 24));
    return a;
}

__global__ void probe_layout(const uint64_t* state /* [64][12] */, uint64_t* out_lo /* [64][12] */, uint64_t* out_hi) {
    const unsigned lane = threadIdx.x;
    uint64_t s[12];
    for (int j = 0; j < 12; j++) s[j] = state[lane * 12 + j];
    const v4i A = weights_for_lane(lane);
    uint64_t lo[12], hi[12];
    for (int i = 0; i < 12; i++) lo[i] = hi[i] = 0;
    for (int b = 0; b < 8; b++) {
        uint8_t pl[16];
        for (int j = 0; j < 16; j++) pl[j] = j < 12 ? (uint8_t)((s[j] >> (8 * b)) ^ 0x80u) : 0;   // signed byte = unsigned - 128
        v4i B;
        for (int d = 0; d < 4; d++) B[d] = (int)(pl[4 * d] | (pl[4 * d + 1] << 8) | (pl[4 * d + 2] << 16) | ((uint32_t)pl[4 * d + 3] << 24));
        v16i C;
        for (int i = 0; i < 16; i++) C[i] = 0;
        const v16i D = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, C, 0, 0, 0);
        for (int i = 0; i < 12; i++) {
            int rowsum = 0;
            for (int j = 0; j < 12; j++) rowsum += mds(i, j);
            const uint64_t sb = (uint64_t)(int64_t)(D[i] + 128 * rowsum);   // back to the sum over unsigned bytes
            if (b < 4) lo[i] += sb << (8 * b);
            else hi[i] += sb << (8 * (b - 4));
        }
    }
    for (int i = 0; i < 12; i++) {
        out_lo[lane * 12 + i] = lo[i];
        out_hi[lane * 12 + i] = hi[i];
    }
}

// timing: ROUNDS x (NM MFMAs interleaved with NV independent v_mad_u64_u32 chains)
template <int NM>
__global__ __launch_bounds__(256, 2) void probe_issue(uint64_t* sink, int rounds, long long* cycles) {
    const unsigned lane = threadIdx.x & 63u;
    v4i A = weights_for_lane(lane), B = {(int)lane, 3, 5, 7};
    v16i C0, C1;
    for (int i = 0; i < 16; i++) C0[i] = C1[i] = 0;
    uint64_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane + i;
    uint32_t x = lane * 2654435761u + 1, y = lane + 77;
    const long long t0 = clock64();
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m < NM) {
                if (m & 1) C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, C1, 0, 0, 0);
                else C0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, C0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 24; k++) acc[k & 7] += (uint64_t)x * (y + k);   // v_mad_u64_u32 with a 64-bit addend
            asm volatile("" : "+v"(x), "+v"(y));
        }
    }
    const long long t1 = clock64();
    uint64_t sum = 0;
    for (int i = 0; i < 8; i++) sum += acc[i];
    for (int i = 0; i < 16; i++) sum += (uint64_t)(uint32_t)(C0[i] + C1[i]);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    uint64_t h_state[64 * 12], *d_state, *d_lo, *d_hi, h_lo[64 * 12], h_hi[64 * 12];
    srand(7);
    for (auto& v : h_state) v = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand() ^ ((uint64_t)rand() << 60);
    for (int j = 0; j < 12; j++) h_state[j] = ~0ull;   // lane 0: every byte 0xff
    for (int j = 0; j < 12; j++) h_state[12 + j] = 0;  // lane 1: zeros
    hipMalloc(&d_state, sizeof h_state); hipMalloc(&d_lo, sizeof h_lo); hipMalloc(&d_hi, sizeof h_hi);
    hipMemcpy(d_state, h_state, sizeof h_state, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_layout, dim3(1), dim3(64), 0, 0, d_state, d_lo, d_hi);
    hipMemcpy(h_lo, d_lo, sizeof h_lo, hipMemcpyDeviceToHost);
    hipMemcpy(h_hi, d_hi, sizeof h_hi, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 12; i++) {
            unsigned __int128 want = 0;
            for (int j = 0; j < 12; j++) want += (unsigned __int128)h_state[l * 12 + j] * (unsigned)mds(i, j);
            const unsigned __int128 got = (unsigned __int128)h_lo[l * 12 + i] + ((unsigned __int128)h_hi[l * 12 + i] << 32);
            if (want != got) {
                if (bad < 8) printf("mismatch lane %d out %d\n", l, i);
                bad++;
            }
        }
    printf("layout: %d mismatches of %d\n", bad, 64 * 12);
    uint64_t* d_sink; long long* d_cyc; long long cyc;
    hipMalloc(&d_sink, 2048 * 256 * 8); hipMalloc(&d_cyc, 8);
    const int rounds = 2000;
    for (int blocks : {1, 2048}) {
        for (int nm : {0, 8}) {
            for (int rep = 0; rep < 2; rep++) {
                if (nm == 0) hipLaunchKernelGGL(probe_issue<0>, dim3(blocks), dim3(256), 0, 0, d_sink, rounds, d_cyc);
                else hipLaunchKernelGGL(probe_issue<8>, dim3(blocks), dim3(256), 0, 0, d_sink, rounds, d_cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
            printf("blocks %4d: %d MFMA + 192 v_mad_u64_u32 per round: %.1f cycles per round (wave 0 of block 0)\n", blocks, nm, (double)cyc / rounds);
        }
    }
    return bad != 0;
}
