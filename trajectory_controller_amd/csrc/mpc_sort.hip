// Longest-first ordering of the projected-gradient work queue.
//
// Iteration counts differ ~30x between instances and are predicted almost perfectly by lambda, the
// Hessian trace bound dlib already computes (Spearman 0.97 with the iteration count on the
// reference's input distribution): the step is 1/lambda, so a large lambda
// means many small steps.  The CD kernel emits a 32-bit key per instance (the bits of
// float(lambda), 0 for instances that already stopped); the PG kernel pulls instances in
// descending key order, which is the LPT rule for the makespan of its persistent lanes.
//
// A counting sort on the top 16 key bits (sign, exponent, 7 mantissa bits: 0.8 % resolution):
// histogram (by the key producer) -> exclusive scan from the largest bin down -> scatter.  Positions inside a bin come
// from an atomic and are therefore not reproducible; they only decide which lane solves which
// instance, never a result.  Integer/HBM-bound, three tiny launches (~15 us at n = 262144).
#include "mpc_internal.h"

namespace tpc {

namespace {

constexpr int kBins = 1 << 16;

// Exclusive scan of the histogram from the largest bin down, in two launches of 64 small blocks:
// a 1024-thread block would have to wait for a CU with sixteen free wave slots, and with a second
// batch in flight (bench.py) every CU is busy with that batch's persistent waves -- the single big
// block used to sit in the queue for milliseconds.  Block b owns the 1024 bins
// [kBins-1-1024b, kBins-1024(b+1)], thread t four consecutive ones of them.
constexpr int kScanBlocks = 64, kScanThreads = 256, kPerThread = kBins / kScanBlocks / kScanThreads;

__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(const uint32_t* __restrict__ hist,
                                                                 uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t part[kScanThreads / 64];
    const int t = threadIdx.x;
    const int top = kBins - 1 - (blockIdx.x * kScanThreads + t) * kPerThread;
    uint32_t sum = 0;
    for (int j = 0; j < kPerThread; ++j) sum += hist[top - j];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((t & 63) == 0) part[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) tot += part[w];
        block_sum[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(uint32_t* __restrict__ hist,
                                                                  const uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t part[kScanThreads];
    const int t = threadIdx.x;
    const int top = kBins - 1 - (blockIdx.x * kScanThreads + t) * kPerThread;
    uint32_t c[kPerThread], sum = 0;
    for (int j = 0; j < kPerThread; ++j) { c[j] = hist[top - j]; sum += c[j]; }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {   // Hillis-Steele inclusive scan
        const uint32_t add = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t base = part[t] - sum;                       // exclusive inside the block
    for (int b = 0; b < (int)blockIdx.x; ++b) base += block_sum[b];
    for (int j = 0; j < kPerThread; ++j) {
        hist[top - j] = base;                            // becomes the bin's first position
        base += c[j];
    }
}

// rank[i] = position of instance i inside its bin (handed out by the key producer together with
// the histogram count), so the scatter needs no atomics: one coalesced read, one 4-byte write.
__global__ void scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ rank,
                               const uint32_t* __restrict__ base, uint32_t* __restrict__ order, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        order[base[keys[i] >> 16] + rank[i]] = (uint32_t)i;
}

}  // namespace

size_t sort_temp_bytes(int64_t) { return (size_t)(kBins + kScanBlocks) * sizeof(uint32_t); }

// The histogram of (key >> 16) is built by the kernel that produces the keys: one atomic per
// instance at the end of the coordinate-descent kernel, whose return value is the instance's rank
// inside its bin.  Spread over that kernel's run time the atomics cost nothing, whereas two separate
// passes of 262 144 atomics piling up on about a thousand hot bins took 64 us + 66 us.
// After order_finish: the number of queue entries above the lowest 128 bins, i.e. where the class
// of instances that need no projected-gradient phase begins (see lane_cd_kernel).
const uint32_t* order_queue_len(const void* temp) { return (const uint32_t*)temp + 127; }

hipError_t order_begin(void* temp, hipStream_t s) {
    return hipMemsetAsync(temp, 0, (size_t)kBins * sizeof(uint32_t), s);
}

// keys[n], rank[n] + histogram in temp -> order[n]: instance indices by descending key.
hipError_t order_finish(const uint32_t* keys, const uint32_t* rank, uint32_t* order, int64_t n, void* temp,
                        hipStream_t s) {
    uint32_t* hist = (uint32_t*)temp;
    const int block = 256;
    int64_t g = (n + block - 1) / block;
    const unsigned grid = (unsigned)(g < 2048 ? g : 2048);
    uint32_t* block_sum = hist + kBins;
    hipLaunchKernelGGL(scan_sums_kernel, dim3(kScanBlocks), dim3(kScanThreads), 0, s, (const uint32_t*)hist, block_sum);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(kScanBlocks), dim3(kScanThreads), 0, s, hist, (const uint32_t*)block_sum);
    hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(block), 0, s, keys, rank, (const uint32_t*)hist, order, n);
    return hipGetLastError();
}

// AUTO's presolve (tpc_mpc_api.cpp): the bit-exact results of the queued instances, computed into side arrays beside the
// tolerance family's pass, over that family's answers
template <typename T>
__global__ void presolve_merge_kernel(const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queue_len,
                                      const T* __restrict__ sf, const T* __restrict__ sr, const int32_t* __restrict__ si,
                                      T* __restrict__ f, T* __restrict__ r, int32_t* __restrict__ it, uint32_t limit) {
    const uint32_t nq = *queue_len;
    if (nq > limit) return;   // the presolve kernel left a queue this long alone (mpc_lanex.h, SOLO)
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const uint32_t k = queue[q];
        f[k] = sf[k];
        r[k] = sr[k];
        if (it) it[k] = si[k];
    }
}
// ... and for the general form: `rows` output rows of leading dimension ld (u0)
__global__ void presolve_merge_rows_kernel(const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queue_len,
                                           const double* __restrict__ su, const int32_t* __restrict__ si, double* __restrict__ u,
                                           int32_t* __restrict__ it, int rows, int64_t ld, uint32_t limit) {
    const uint32_t nq = *queue_len;
    if (nq > limit) return;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const uint32_t k = queue[q];
        for (int j = 0; j < rows; ++j) u[(int64_t)j * ld + k] = su[(int64_t)j * ld + k];
        if (it) it[k] = si[k];
    }
}
hipError_t presolve_merge_rows(const uint32_t* queue, const uint32_t* queue_len, int64_t n, const void* su, const int32_t* si,
                               void* u, int32_t* it, int rows, int64_t ld, uint32_t limit, hipStream_t s) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(presolve_merge_rows_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s, queue, queue_len,
                       (const double*)su, si, (double*)u, it, rows, ld, limit);
    return hipGetLastError();
}
hipError_t presolve_merge(const uint32_t* queue, const uint32_t* queue_len, int64_t n, const void* sf, const void* sr,
                          const int32_t* si, void* f, void* r, int32_t* it, uint32_t limit, hipStream_t s) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(presolve_merge_kernel<double>, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s, queue, queue_len,
                       (const double*)sf, (const double*)sr, si, (double*)f, (double*)r, it, limit);
    return hipGetLastError();
}

}  // namespace tpc
