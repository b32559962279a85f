// spp_dense_444.h -- the bulk trailing-update tile built around v_mfma_f64_4x4x4_4b (included by spp_dense.hip).
//
// Measured on MI355X (tools/mfma_rate.hip): v_mfma_f64_16x16x4 issues once per ~101 cycles and SIMD (48 TFLOP/s chip-wide),
// v_mfma_f64_4x4x4_4b -- four independent 4 x 4 x 4 products, 512 flop -- once per 18 cycles (72 TFLOP/s). The small form
// needs four times the operand words per flop, so it only pays with the operands held in registers across an outer
// product and NO staging registers: the round-2 macro swap inside the 16x16x4 kernel (operands through 16 staging VGPRs,
// 2 x 2 register tiles) spilled at 64 VGPRs and reached 20-30 TFLOP/s. This kernel is built for the instruction:
//   * C tile 128 x 128 per 1024-thread workgroup, 64 VGPRs, two workgroups per CU (as before); a wave owns 32 x 32 of it
//     as 8 x 2 accumulators: with the first operand replicated over its four blocks one instruction is a
//     (4 columns of C) x (16 rows of C) outer-product step over 4 k: acc[g][h] <- C(m = 16 h + (lane & 15), n = 4 g + (lane >> 4)),
//     so lanes 0..15 of every accumulator still hold 16 consecutive rows of a column of C (128-byte segments);
//   * per 4 k: 2 + 8 operand words from LDS feed 16 instructions (288 cycles of matrix pipe per wave);
//   * the row panels go from L2 / HBM straight into LDS (global_load_lds_dwordx4, 16-deep slabs, two buffers: the next slab
//     lands while the current one is multiplied) -- no staging registers, no ds_write;
//   * LDS image of a slab: column slot q at 128 bytes x q, NO padding. Conflict-free ds_read_b64 fragments come from the
//     placement instead -- which the DMA leaves free, every lane names its own source address: column m of a group of
//     16 sits in slot ((m & 7) << 1) | (m >> 3), its 16-byte pieces are XOR-swizzled with (m & 7). The 32 lanes of one
//     LDS cycle (16 columns x 2 k) then hit 32 different bank pairs.
#pragma once

namespace spp {

constexpr int T444_SLAB = 16;                        // k-depth of a slab
constexpr int T444_IMG = 128 * T444_SLAB;            // doubles per panel image
constexpr int T444_LDS_DOUBLES = 2 * 2 * T444_IMG;   // two buffers x (A image + B image) = 64 KB

// LDS offset (doubles) of element (column c of the 128, k of the slab)
__device__ __forceinline__ int t444_off(int c, int k)
{
	const int slot = (c & ~15) | ((c & 7) << 1) | ((c >> 3) & 1);
	return slot * T444_SLAB + 2 * ((k >> 1) ^ (c & 7)) + (k & 1);
}

// INTERIOR tiles only: rows [m0, m0 + 128) and columns [n0, n0 + 128) lie inside the matrix (the caller sends the edge
// tiles through the bounds-checked 16x16x4 tile); K a multiple of 16
// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() would also wait for the slab the DMA has
// just been asked for (vmcnt(0)) and the prefetch would overlap nothing
__device__ __forceinline__ void t444_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int SC1 = 0>
__device__ __forceinline__ void gemm_tn_tile_444(const int64_t m0, const int64_t n0, const int K,
	const double *__restrict__ A, const int64_t lda, const double *B, const int64_t ldb, double *C, const int64_t ldc, double *lds)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave & 3) * 32, wn = (wave >> 2) * 32; // wave-uniform corner of the wave's 32 x 32 part
	const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3;
	// ---- DMA source of this lane: wave w fills the slots 8 w .. 8 w + 7 of both images; lane -> (slot, 16-byte piece)
	const double *ga, *gb;
	{
		const int q = wave * 8 + (lane >> 3), pp = lane & 7;
		const int col = (q & ~15) | ((q & 1) << 3) | ((q >> 1) & 7); // the column that lives in slot q
		const int p = pp ^ (col & 7);                                // the piece of it that lives at position pp
		ga = A + (m0 + col) * lda + 2 * p;
		gb = B + (n0 + col) * ldb + 2 * p;
	}
	double *const img = lds + wave * 128; // this wave's kilobyte of an image
	auto issue = [&](const int s, const int buf) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + T444_SLAB * s),
			(__attribute__((address_space(3))) void*)(img + buf * 2 * T444_IMG), 16, 0, 0);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + T444_SLAB * s),
			(__attribute__((address_space(3))) void*)(img + buf * 2 * T444_IMG + T444_IMG), 16, 0, 0);
	};
	issue(0, 0);
	// ---- the old C values, negated, are the initial accumulators: acc = -C, C_new = -(acc + A^T B)
	double acc[8][2];
	const uint32_t c_lane = (uint32_t)(l15 + l4 * ldc);
#pragma unroll
	for(int g = 0; g < 8; ++ g)
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			const int64_t mu = m0 + wm + 16 * h, nu = n0 + wn + 4 * g;
			acc[g][h] = -(C + mu + nu * ldc)[c_lane];
		}
	// per-lane parts of the operand addresses; everything else is a compile-time offset of the ds_read:
	//   m side, column wm + 16 h + l15:  slot * 16 = 16 wm + 256 h + 16 pi(l15),  swizzle 2 ((k4 >> 1) ^ tm), tm = (l4 >> 1) ^ (l15 & 7)
	//   n side, column wn + 4 g + l3:    slot * 16 = 16 wn + 256 (g >> 2) + 128 (g & 1) + 16 ((g >> 1) & 1) + 32 l3,
	//                                    swizzle 2 ((k4 >> 1) ^ 4 (g & 1) ^ tn), tn = (l4 >> 1) ^ l3
	// (k = k4 + l4 and k4 is a multiple of 4: (k >> 1) = (k4 >> 1) | (l4 >> 1), whose bits XOR independently)
	const int pm = 16 * wm + 16 * (((l15 & 7) << 1) | (l15 >> 3)) + (l4 & 1), tm = (l4 >> 1) ^ (l15 & 7);
	const int pn = 16 * wn + 32 * l3 + (l4 & 1), tn = (l4 >> 1) ^ l3;
	const int ns = K / T444_SLAB;
	for(int s = 0; s < ns; ++ s) {
		if(s + 1 < ns) {
			issue(s + 1, (s + 1) & 1);
			asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // everything but the two loads just issued has landed
		} else
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		t444_barrier(); // every wave's part of slab s has landed
		const double *Ai = lds + (s & 1) * 2 * T444_IMG, *Bi = Ai + T444_IMG;
#pragma unroll
		for(int kh = 0; kh < T444_SLAB / 4; ++ kh) { // k4 = 4 kh, k4 >> 1 = 2 kh
			const double *am = Ai + pm + 2 * ((2 * kh) ^ tm);
			const double *an0 = Bi + pn + 2 * ((2 * kh) ^ tn), *an1 = Bi + pn + 2 * ((2 * kh) ^ 4 ^ tn);
			const double fm0 = am[0], fm1 = am[256];
#pragma unroll
			for(int half = 0; half < 2; ++ half) {
				double fn[4];
#pragma unroll
				for(int j = 0; j < 4; ++ j) {
					const int g = 4 * half + j;
					fn[j] = ((g & 1) ? an1 : an0)[128 * (g & 1) + 16 * ((g >> 1) & 1) + 256 * (g >> 2)];
				}
#pragma unroll
				for(int j = 0; j < 4; ++ j) {
					const int g = 4 * half + j;
					acc[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(fn[j], fm0, acc[g][0], 0, 0, 0);
					acc[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(fn[j], fm1, acc[g][1], 0, 0, 0);
				}
				// (without this fence the scheduler hoists the next half's operand loads over the instructions above and the
				// register allocator spills 50 VGPRs at the 64 the two-workgroups-per-CU budget allows)
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		t444_barrier(); // the buffer is refilled by the DMA of the next iteration
	}
#pragma unroll
	for(int g = 0; g < 8; ++ g)
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			const int64_t mu = m0 + wm + 16 * h, nu = n0 + wn + 4 * g;
			if(SC1)
				__hip_atomic_store(&(C + mu + nu * ldc)[c_lane], -acc[g][h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else
				(C + mu + nu * ldc)[c_lane] = -acc[g][h];
		}
}

// The same staging (LDS-DMA, two 16-deep slab buffers, swizzled unpadded images, no staging registers) under the
// v_mfma_f64_16x16x4 instruction: a wave's 32 x 32 part as 2 x 2 accumulators of four doubles, 4 operand words per 4 k for
// 4 instructions. Measured with the operand pattern of a real tile product -- a register outer product, distinct
// operand registers per instruction (tools/mfma_rate2.hip) -- this form issues once per 75 cycles and SIMD = 67.5 TFLOP/s,
// the 4x4x4_4b form once per 16-17 cycles = 76.7 TFLOP/s: 14 % apart, not the 48 against 72 that the loop with ONE
// operand pair for every instruction (tools/mfma_rate.hip) had shown. The small form's fourfold operand traffic costs
// more than that (35.9 against 39.6-41 TFLOP/s stand-alone for the two tiles), so the large form is the one in use.
typedef double t444_v4 __attribute__((ext_vector_type(4)));

template <int SC1 = 0>
__device__ __forceinline__ void gemm_tn_tile_dma(const int64_t m0, const int64_t n0, const int K,
	const double *__restrict__ A, const int64_t lda, const double *B, const int64_t ldb, double *C, const int64_t ldc, double *lds)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave & 3) * 32, wn = (wave >> 2) * 32;
	const int l15 = lane & 15, l4 = lane >> 4;
	const double *ga, *gb;
	{
		const int q = wave * 8 + (lane >> 3), pp = lane & 7;
		const int col = (q & ~15) | ((q & 1) << 3) | ((q >> 1) & 7);
		const int p = pp ^ (col & 7);
		ga = A + (m0 + col) * lda + 2 * p;
		gb = B + (n0 + col) * ldb + 2 * p;
	}
	double *const img = lds + wave * 128;
	auto issue = [&](const int s, const int buf) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + T444_SLAB * s),
			(__attribute__((address_space(3))) void*)(img + buf * 2 * T444_IMG), 16, 0, 0);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + T444_SLAB * s),
			(__attribute__((address_space(3))) void*)(img + buf * 2 * T444_IMG + T444_IMG), 16, 0, 0);
	};
	issue(0, 0);
	// D[row = l4 + 4 r][col = l15] of accumulator (g, h) is C(m = wm + 16 h + l15, n = wn + 16 g + l4 + 4 r)
	t444_v4 acc[2][2];
	const uint32_t c_lane = (uint32_t)(l15 + l4 * ldc);
#pragma unroll
	for(int g = 0; g < 2; ++ g)
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			const double *Cu = C + (m0 + wm + 16 * h) + (n0 + wn + 16 * g) * ldc;
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				acc[g][h][r] = -(Cu + (int64_t)(4 * r) * ldc)[c_lane];
		}
	// both sides read the natural fragment: column (wave part) + 16 x + l15, k = k4 + l4 (t444_off, split into the
	// per-lane part and compile-time offsets as in the 4x4x4 tile)
	const int pl = 16 * (((l15 & 7) << 1) | (l15 >> 3)) + (l4 & 1), tm = (l4 >> 1) ^ (l15 & 7);
	const int ns = K / T444_SLAB;
	for(int s = 0; s < ns; ++ s) {
		if(s + 1 < ns) {
			issue(s + 1, (s + 1) & 1);
			asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
		} else
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		t444_barrier();
		const double *Ai = lds + (s & 1) * 2 * T444_IMG + 16 * wm + pl, *Bi = lds + (s & 1) * 2 * T444_IMG + T444_IMG + 16 * wn + pl;
#pragma unroll
		for(int kh = 0; kh < T444_SLAB / 4; ++ kh) {
			const int sw = 2 * ((2 * kh) ^ tm);
			const double fm0 = Ai[sw], fm1 = Ai[sw + 256], fn0 = Bi[sw], fn1 = Bi[sw + 256];
			acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn0, fm0, acc[0][0], 0, 0, 0);
			acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn0, fm1, acc[0][1], 0, 0, 0);
			acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn1, fm0, acc[1][0], 0, 0, 0);
			acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn1, fm1, acc[1][1], 0, 0, 0);
		}
		t444_barrier();
	}
#pragma unroll
	for(int g = 0; g < 2; ++ g)
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			double *Cu = C + (m0 + wm + 16 * h) + (n0 + wn + 16 * g) * ldc;
#pragma unroll
			for(int r = 0; r < 4; ++ r) {
				if(SC1)
					__hip_atomic_store(&(Cu + (int64_t)(4 * r) * ldc)[c_lane], -acc[g][h][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				else
					(Cu + (int64_t)(4 * r) * ldc)[c_lane] = -acc[g][h][r];
			}
		}
}

} // namespace spp
