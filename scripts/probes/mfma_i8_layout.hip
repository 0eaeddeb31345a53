// Probe: operand / result lane maps of v_mfma_i32_16x16x64_i8 on gfx950, with exact integer data.
// A[r][k] = r + 1 (k-independent part) * indicator(k == K0); B[k][c] = (c + 1) * indicator(k == K1) ...
// Simpler: fill A, B with small random ints on the host in the ASSUMED layout, compare D with a host GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i* a, const v4i* b, v4i* d) {
    int l = threadIdx.x;
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}
int main() {
    int8_t A[16][64], B[64][16];
    srand(1);
    for (int r = 0; r < 16; r++) for (int k = 0; k < 64; k++) A[r][k] = rand() % 7 - 3;
    for (int k = 0; k < 64; k++) for (int c = 0; c < 16; c++) B[k][c] = rand() % 255 - 127;
    int32_t D[16][16];
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { int s = 0; for (int k = 0; k < 64; k++) s += A[r][k] * B[k][c]; D[r][c] = s; }
    // assumed: lane l: A[row l&15][k = 16*(l>>4) + j], j = byte index 0..15 (VGPR j/4, byte j%4); same for B cols
    int8_t ha[64][16], hb[64][16];
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) { ha[l][j] = A[l & 15][16 * (l >> 4) + j]; hb[l][j] = B[16 * (l >> 4) + j][l & 15]; }
    v4i *da, *db, *dd; int32_t hd[64][4];
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int reg = 0; reg < 4; reg++) {
        int col = l & 15, row = (l >> 4) * 4 + reg;
        if (hd[l][reg] != D[row][col]) bad++;
    }
    printf("assumed layout (A,B: k = 16*(l>>4)+j ; D: col=l&15,row=4*(l>>4)+reg): %d mismatches of 256\n", bad);
    return bad != 0;
}
