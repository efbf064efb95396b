// Internal C++ interfaces shared between the translation units of libppk.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ppk {

// C[M,N] = alpha * opA(A) . opB(B) + beta * C + bias[N]   (row-major, fp32 MFMA)
//   transA == 0: A is [M,K] (lda >= K)      transA != 0: A is [K,M] (lda >= M)
//   transB == 0: B is [K,N] (ldb >= N)      transB != 0: B is [N,K] (ldb >= K)
int gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
         const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
         float* C, int64_t ldc, const float* bias, void* ws, size_t ws_bytes, hipStream_t stream);
size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

}  // namespace ppk
