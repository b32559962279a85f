// spp_dense_tail.h -- the tail of the dense factor (trailing matrix of at most ~20 tile rows) as ONE launch whose
// workgroups own one 128 x 128 tile each and stream the factorization through it 16 rows at a time.
// (included by spp_dense.hip; device code + the argument block)
//
// The single-stream tail of the round-2 schedule costs 42 us per step: fused [update + potrf_diag] 31, panel solve 8,
// launch gaps 3 -- and potrf_diag, row panel and update of one step follow each other. Here
//   * tile (i, j), i <= j, of the tail lives in the LDS / registers of workgroup (i, j) from the start of the launch
//     until its own step i (20 tile rows = 210-230 tiles: one workgroup per CU, all resident);
//   * the workgroup of a diagonal tile factors it with potrf_diag_body<HALF = 2> (no inverse inside the loop), which
//     publishes row tile J of R_ii and the inverse of its 16 x 16 diagonal tile as soon as panel J is done;
//   * the workgroups (i, j > i) of the row panel consume them as they appear: X_J = Dinv_J^T Y_J, Y_I -= R_JI^T X_J
//     (substitution by 16-row tiles, no block inverse), publish X_J = rows 16 J .. 16 J + 15 of R(i, j);
//   * every workgroup below, (i', j') with i' > i, applies the rank-16 update  T -= R(i, i')_J^T R(i, j')_J  to its
//     accumulators the moment both row tiles are out -- so tile (i+1, i+1) has the whole update of step i a few
//     microseconds after potrf_diag(i) ends, already sits in the LDS of the workgroup that factors it, and that
//     workgroup starts at once: a step costs potrf_diag's panel loop (19 us) plus one hand-over chain, not 42 us.
// The inverse of each diagonal block (the backward substitution wants it) is formed by the same workgroup AFTER its
// factorization, off the chain. Hand-overs: the payload is stored write-through (agent-scope atomic stores), every storing
// wave drains (s_waitcnt vmcnt(0)), workgroup barrier, one lane stores the counter; a consumer polls the counter with
// one lane (relaxed, agent scope) and reads the payload with agent-scope atomic loads (L2-bypassing on gfx950) -- no
// fences. Counters carry the epoch of the factorization in their upper bits (no clearing between factorizations).
// Workgroups are numbered row by row: whatever a workgroup waits for is produced by a workgroup
// with a smaller index, which never waits for a larger one -- progress does not depend on residency. Every wait is bounded (abort word).
// Summation order per tile: steps ascending, row tiles ascending -- fixed, bit-reproducible.
#pragma once
#include "spp_dense_dev.h"

namespace spp {

struct TailArgs {
	double *A;           // the matrix (column-major, leading dimension ld)
	int64_t ld;
	int64_t rows;        // pivot rows of the matrix (rows >= this are identity padding)
	int64_t ncols;       // columns (the right-hand side column included)
	int64_t c0;          // first row / column of the tail region
	int have_pre;        // 1: the row panel of the step before the region (rows c0 - 128 .. c0 - 1) is complete in memory and not yet applied
	int Tr, Tc;          // tile rows / tile columns of the region
	int has_rhs;
	double *tinv;        // block inverses of the region's diagonal blocks (Tr x 128 x 128)
	double *dbuf;        // [Tr][8][256]: inverse 16 x 16 diagonal tiles as they are published
	int *pub;            // [Tr][Tc]: (epoch << 4) | row tiles published
	int epoch;
	int *info, *abort;
	long long timeout_ticks;
	long long *trace;    // debugging (SPP_TAIL_TRACE): per tile row 8 wall-clock stamps
	const int *order;    // workgroup -> tile (i << 16 | j), a topological order of the tiles (see the host side)
};

constexpr int TAIL_LDS_DOUBLES = POTRF_LDS_DOUBLES_INV2 + 16;

// A published row tile is read with agent-scope atomic loads (they bypass the L2s). Plain loads would be safe as well -- a
// 128-byte line is exactly one row tile's piece of one column, written once, write-through, before the counter that
// announces it, and untouched by anybody but its owner before -- and would let the 20-40 workgroups that need the same
// row tile share it through their XCD's L2; measured slower (SPP_TAIL_PLAIN_LD=1: factor 2.54 -> 2.94 ms).
// (The rank-16 updates with v_mfma_f64_4x4x4_4b -- the A operand replicated over its four blocks makes an instruction a
// (4 x 4)(4 x 16) product whose strip lies in accumulator register ib exactly like in the 16x16x4 form, the B fragment is
// the same, the A fragment is element (k = 4 kc + l4, row 4 ib + (lane & 3)) -- were built, parity-green, and measured
// slower: factor 2.45 -> 3.27 ms; the update is bound by its LDS reads, and that form needs 16 A fragments per tile.)
#ifndef SPP_TAIL_PLAIN_LD
#define SPP_TAIL_PLAIN_LD 0
#endif
__device__ __forceinline__ double tail_ld(const double *p)
{
#if SPP_TAIL_PLAIN_LD
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (an ordinary cached load the compiler may not move)
#else
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// (a function of its own: inlined into the update loop its registers add to the loop's and the loop spills)
__device__ __noinline__ void tail_first_tile(double *sm, const v4f64 t00, const int lane, int *info, const int64_t k0)
{
	const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
	for(int r = 0; r < 4; ++ r)
		sm[(l4 + 4 * r) + l15 * TS] = t00[r];
	if(lane == 0)
		potrf_lds2_fail(sm)[0] = 0;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	diag_tile_factor<TS>(sm, potrf_lds2_dv(sm), potrf_lds2_gd0(sm), potrf_lds2_dinv(sm), 0, lane, potrf_lds2_fail(sm), info, k0);
}

__global__ __launch_bounds__(POTRF_THREADS)
void dense_tail_kernel(const TailArgs a)
{
	extern __shared__ double sm[];
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l15 = lane & 15, l4 = lane >> 4;
	int *st = (int*)(sm + POTRF_LDS_DOUBLES_INV2); // st[0] = wait succeeded, st[1] = row tiles available
	// ---- which tile: the upper tiles enumerated row by row. Tile (i, j) needs row tiles of (k, i) and (k, j), k < i:
	// earlier rows, smaller indices. With more tiles than CUs (the whole Venice factorization: 861) the first ~6 rows are
	// resident at the start, a row's workgroups leave after its step and later rows move in, finding the row tiles of the
	// steps they missed in memory (they catch up at the speed of their matrix cores: 2 us per row tile against the 3 us
	// at which the chain emits them). Two other topological orders were measured on the whole factorization (kernel
	// time, row by row 2.05 ms): by anti-diagonals (i + j: a deadline order for the diagonal -- but the tiles of one
	// column then run one after the other, each redoing all its updates in one go: 2.33 ms) and column by column (the
	// first 20 steps at 41 us each, then every late column redoes up to 35 steps of updates at once: 3.05 ms).
	const int ord = a.order[blockIdx.x];
	const int ti = ord >> 16, tj = ord & 0xffff;
	const bool diag = ti == tj;
	const int64_t i0 = a.c0 + (int64_t)NB * ti, j0 = a.c0 + (int64_t)NB * tj;
	const int tag = a.epoch << 4;
	auto stamp = [&](int slot) { if(a.trace && tid == 0) a.trace[ti * 8 + slot] = wall_clock64(); };
	if(diag) stamp(0);
	// how many row tiles two counters say are out (0 for a counter of another epoch)
	auto pub_count = [&](const int v, const int w) -> int {
		const int cv = ((v >> 4) == a.epoch) ? (v & 15) : 0, cw = ((w >> 4) == a.epoch) ? (w & 15) : 0;
		return cv < cw ? cv : cw;
	};
	// bounded wait by one lane: counters p (and q) of this epoch at least `need`; returns how many row tiles are out
	auto wait_pub = [&](const int *p, const int *q, const int need) -> int {
		if(tid == 0) {
			const long long t0 = wall_clock64();
			int got = -1;
			for(;;) {
				const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const int w = q ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : v;
				const int cv = ((v >> 4) == a.epoch) ? (v & 15) : 0, cw = ((w >> 4) == a.epoch) ? (w & 15) : 0;
				const int c = cv < cw ? cv : cw;
				if(c >= need) {
					got = c;
					break;
				}
				if(__hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
					break;
				if(wall_clock64() - t0 > a.timeout_ticks) {
					__hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			st[1] = got;
		}
		lds_barrier();
		const int got = st[1];
		lds_barrier(); // (st is rewritten by the next wait)
		return got;
	};
	// ---- the tile into accumulators. Wave w holds up to four 16 x 16 tiles (ra[u], cb[u]); element (16 ra + l4 + 4 r,
	// 16 cb + l15) of the tile in acc[u][r] (the MFMA result layout of tile_atb). An off-diagonal tile: (w & 7, 4 (w >> 3) + u);
	// a diagonal tile: the 36 tiles on and above its diagonal dealt round-robin, eight to ten per SIMD -- the rank-16
	// update of a whole tile is 1 - 2 us of one CU's matrix cores, and on a diagonal tile it sits on the critical chain
	int ra[4], cb[4];
#pragma unroll
	for(int u = 0; u < 4; ++ u) {
		if(!diag) {
			ra[u] = wave & 7;
			cb[u] = 4 * (wave >> 3) + u;
		} else {
			// upper tiles enumerated column by column (column c holds c + 1 of them): wave 0 owns the first one ALONE -- it
			// eliminates it under the tile's last update --, waves 1 .. 15 share the other 35
			int q = (wave == 0) ? (u == 0 ? 0 : 36) : wave + 15 * u, c = 0;
			if(q >= 36)
				ra[u] = cb[u] = -1;
			else {
				while(q > c) {
					q -= c + 1;
					++ c;
				}
				ra[u] = q;
				cb[u] = c;
			}
		}
	}
	v4f64 acc[4];
	{
		// through an LDS image: coalesced loads (128 consecutive rows of a column per wave pair), then every lane picks
		// its accumulator elements
		double *img = sm;
#pragma unroll 4
		for(int q = 0; q < NB * NB / POTRF_THREADS; ++ q) {
			const int e = tid + q * POTRF_THREADS, r = e & (NB - 1), c = e >> 7;
			const int64_t row = i0 + r, col = j0 + c;
			// (columns beyond the right-hand side are padding: zero, the identity on a diagonal tile's diagonal)
			img[r + c * TS] = (col < a.ncols) ? a.A[row + col * a.ld] : ((diag && row == col) ? 1.0 : 0.0);
		}
		lds_barrier();
#pragma unroll
		for(int u = 0; u < 4; ++ u)
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				acc[u][r] = (ra[u] < 0) ? 0.0 : img[(16 * ra[u] + l4 + 4 * r) + (16 * cb[u] + l15) * TS];
		lds_barrier();
	}
	bool pre_first = false; // the diagonal tile's first 16 x 16 tile is eliminated under its last update
	// ---- steps before this tile's own: T -= R(k, i)_J^T R(k, j)_J as the row tiles appear.
	// A row tile (16 x 128, column-major) is fetched by the whole workgroup -- 16 consecutive lanes on the 128 bytes of one
	// column, two elements per thread and tile, all loads in flight together -- into an LDS image [k + c * PT]; the MFMA
	// operands come out of LDS. Two images per side: while row tile s is applied, row tile s + 1 -- if it is out already,
	// i.e. whenever this workgroup lags its producers -- is on its way (a per-lane gather of the MFMA fragments straight
	// from memory, 20 loads of 32-byte pieces per wave and row tile, took 6.4 us per row tile against the 2.3 us at
	// which the factorization emits them).
	{
		const int ek = tid & 15, ec = tid >> 4; // this thread's elements: (ek, ec) and (ek, ec + 64) of a row tile
		constexpr int PS = 18;                  // column stride of a row tile's LDS image (even: 16-byte aligned columns)
		const int ekp = 4 * (ek & 3) + (ek >> 2);
		const int kfirst = a.have_pre ? -1 : 0, nst = 8 * (ti - kfirst);
		double va[2], vb[2];
		auto fetch = [&](const int sidx) {
			const int k = kfirst + (sidx >> 3), J = sidx & 7;
			const double *rowp = a.A + (a.c0 + (int64_t)NB * k) + 16 * J + ek;
#pragma unroll
			for(int h = 0; h < 2; ++ h) {
				const int64_t ca = i0 + ec + 64 * h, cb = j0 + ec + 64 * h;
				// (a diagonal tile uses ONE image for both sides: it must hold the right-hand side column too; the rows of the
				// tile beyond the pivots -- identity padding, no update -- are masked when the A fragments are formed)
				va[h] = (ca < (diag ? a.ncols : a.rows)) ? tail_ld(rowp + ca * a.ld) : 0.0;
				vb[h] = (!diag && cb < a.ncols) ? tail_ld(rowp + cb * a.ld) : 0.0;
			}
		};
		int avail = a.have_pre ? 8 : 0; // row tiles of the current step known to be out
		pre_first = diag && nst > 0 && (nst & 1) == 0 && a.rows - i0 >= 16; // (the last round's images are the second pair; the right-hand side column is not in the first tile)
		bool inflight = false;
		for(int sidx = 0; sidx < nst; ++ sidx) {
			const int k = kfirst + (sidx >> 3), J = sidx & 7;
			if(J == 0 && k >= 0)
				avail = 0;
			if(!inflight) {
				if(J >= avail) {
					avail = wait_pub(a.pub + k * a.Tc + ti, a.pub + k * a.Tc + tj, J + 1);
					if(avail < 0)
						return;
				}
				fetch(sidx);
			}
			// a look at the counters for the NEXT row tile of this step, by one lane, without waiting: the answer travels
			// with the barrier below. (A workgroup that keeps up only ever learns of one row tile per wait; one that lags
			// finds the next one out already and fetches it under this one's products -- that is how it catches up.)
			int pk = -1;
			if(tid == 0 && k >= 0 && J + 1 < 8 && J + 1 >= avail)
				pk = pub_count(__hip_atomic_load(a.pub + k * a.Tc + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
					__hip_atomic_load(a.pub + k * a.Tc + tj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			// image of a row tile: element (k, c) at kp(k) + c * PS with kp(k) = 4 (k & 3) + (k >> 2): the four k values a
			// lane feeds to its four MFMAs (k = 4 kk + l4) are then consecutive -- two 16-byte LDS reads per fragment
			// instead of four 8-byte ones (the update is bound by its LDS reads, not by the matrix cores)
			double *sa = sm + (sidx & 1) * 2 * NB * PS, *sb = diag ? sa : sa + NB * PS; // two images per side, alternating
#pragma unroll
			for(int h = 0; h < 2; ++ h) {
				sa[ekp + (ec + 64 * h) * PS] = va[h];
				if(!diag)
					sb[ekp + (ec + 64 * h) * PS] = vb[h];
			}
			if(tid == 0)
				st[2 + (sidx & 1)] = pk;
			lds_barrier();
			{
				const int seen = st[2 + (sidx & 1)];
				if(seen > avail)
					avail = seen;
			}
			// the next row tile, if it is out already (same step; the next step's counters are other words)
			inflight = (sidx + 1 < nst) && (J + 1 < 8) && (J + 1 < avail);
			if(inflight)
				fetch(sidx + 1);
			double fa[4]; // (an off-diagonal tile: the four tiles of a wave share their rows -- ONE A fragment for all four)
#pragma unroll
			for(int u = 0; u < 4; ++ u) {
				if(ra[u] < 0)
					continue;
				double fb[4];
				if(u == 0 || diag) {
					// (the rows of the tile beyond the pivots are identity padding: no update)
					const bool row_live = i0 + 16 * ra[u] + l15 < a.rows;
					const double2 a01 = *(const double2*)(sa + 4 * l4 + (16 * ra[u] + l15) * PS), a23 = *(const double2*)(sa + 4 * l4 + 2 + (16 * ra[u] + l15) * PS);
					fa[0] = row_live ? -a01.x : 0.0;
					fa[1] = row_live ? -a01.y : 0.0;
					fa[2] = row_live ? -a23.x : 0.0;
					fa[3] = row_live ? -a23.y : 0.0;
				}
				{
					const double2 b01 = *(const double2*)(sb + 4 * l4 + (16 * cb[u] + l15) * PS), b23 = *(const double2*)(sb + 4 * l4 + 2 + (16 * cb[u] + l15) * PS);
					fb[0] = b01.x;
					fb[1] = b01.y;
					fb[2] = b23.x;
					fb[3] = b23.y;
				}
#pragma unroll
				for(int kk = 0; kk < 4; ++ kk)
					acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk], fb[kk], acc[u], 0, 0, 0);
				if(u == 0 && pre_first && sidx == nst - 1 && wave == 0) {
					// a diagonal tile's LAST update: wave 0 owns its first 16 x 16 tile and has it complete now. It puts it
					// where the image will have it (that part of LDS held the row-tile images of the round before: free) and
					// eliminates it at once -- 5 500 cycles the other waves spend on their last tiles and on the image --,
					// then takes its other two tiles
					tail_first_tile(sm, acc[0], lane, a.info, i0);
				}
			}
		}
		lds_barrier(); // (the images are overwritten by the tile below)
	}
	if(diag) stamp(1);
	// ---- own step: the accumulators become the LDS image
	double *T = sm;
#pragma unroll
	for(int u = 0; u < 4; ++ u)
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			if(ra[u] >= 0 && !(pre_first && wave == 0 && u == 0))
				T[(16 * ra[u] + l4 + 4 * r) + (16 * cb[u] + l15) * TS] = acc[u][r];
	lds_barrier();
	if(diag) {
		const int64_t nv = a.rows - i0;
		const int n_valid = (int)(nv < 0 ? 0 : (nv > NB ? NB : nv));
		PotrfPub pb;
		pb.flag = a.pub + ti * a.Tc + ti;
		pb.base = tag;
		pb.dbuf = a.dbuf + (size_t)ti * 8 * 256;
		pb.abort = a.abort;
		potrf_diag_body<false, 1, 2, true>(a.A + i0 + i0 * a.ld, a.ld, n_valid, (a.has_rhs && n_valid < NB) ? 1 : 0,
			a.tinv + (size_t)ti * NB * NB, a.info, i0, sm, pb, pre_first ? 1 : 0);
		stamp(2);
		return;
	}
	// ---- a tile of the row panel of step ti: substitution by 16-row tiles behind the factorization of (ti, ti)
	double *Dv = T + NB * TS;       // Dinv_J[k][i] at k + i * PT
	double *Rr = Dv + 16 * PT;      // row tile J of R_ii right of its diagonal tile: element (k, c) at k + c * PT, c = column - 16 (J + 1)
	const double *Rii = a.A + i0 + i0 * a.ld;
	int avail = 0;
	bool inflight = false;
	double dvr = 0, rr[2] = {0, 0}; // this thread's pieces of Dinv_J and of the row tile, fetched one row tile ahead when it is out already
	auto fetch_panel = [&](const int J) {
		if(tid < 256)
			dvr = tail_ld(a.dbuf + ((size_t)ti * 8 + J) * 256 + tid);
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			const int e = tid + POTRF_THREADS * h, k = e & 15, c = e >> 4;
			if(e < 16 * (NB - 16 * (J + 1)))
				rr[h] = tail_ld(Rii + (16 * J + k) + (int64_t)(16 * (J + 1) + c) * a.ld);
		}
	};
	for(int J = 0; J < 8; ++ J) {
		if(!inflight) {
			if(J >= avail) {
				avail = wait_pub(a.pub + ti * a.Tc + ti, nullptr, J + 1);
				if(avail < 0)
					return;
			}
			fetch_panel(J);
		}
		if(tj == ti + 1 && J == 0) stamp(3);
		if(tj == ti + 1 && J == 7) stamp(4);
		int pk = -1; // (a look at the counter for the next row tile, as above)
		if(tid == 0 && J + 1 < 8 && J + 1 >= avail) {
			const int v = __hip_atomic_load(a.pub + ti * a.Tc + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			pk = pub_count(v, v);
		}
		// stage Dinv_J and the row tile
		if(tid < 256)
			Dv[(tid & 15) + (tid >> 4) * PT] = dvr;
#pragma unroll
		for(int h = 0; h < 2; ++ h) {
			const int e = tid + POTRF_THREADS * h;
			if(e < 16 * (NB - 16 * (J + 1)))
				Rr[(e & 15) + (e >> 4) * PT] = rr[h];
		}
		if(tid == 0)
			st[2 + (J & 1)] = pk;
		lds_barrier();
		{
			const int seen = st[2 + (J & 1)];
			if(seen > avail)
				avail = seen;
		}
		inflight = (J + 1 < 8) && (J + 1 < avail);
		if(inflight)
			fetch_panel(J + 1);
		// X_J = Dinv_J^T Y_J: waves 0 .. 7, one column tile each; stored (write-through) as rows 16 J .. of R(ti, tj)
		if(wave < 8) {
			double *Y = T + 16 * J + (16 * wave) * TS;
			const v4f64 x = tile_atb(Dv, 1, PT, Y, 1, TS, lane);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			const int64_t col = j0 + 16 * wave + l15;
#pragma unroll
			for(int r = 0; r < 4; ++ r) {
				Y[(l4 + 4 * r) + l15 * TS] = x[r];
				if(col < a.ncols)
					__hip_atomic_store(&a.A[(i0 + 16 * J + l4 + 4 * r) + col * a.ld], x[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		lds_barrier();
		// Y_I -= R_JI^T X_J for the row tiles I > J. A wave always works on column tile b = wave & 7: its X_J fragment is
		// fetched from LDS once. FIRST the two row tiles the next two substitution steps need (I = J + 1 by waves 0 .. 7,
		// I = J + 2 by waves 8 .. 15); then the row tile goes out (the storing waves' stores have had that time; drain,
		// barrier, one counter store); the REST of the update (I > J + 2) comes after that, in the time this workgroup would
		// otherwise spend waiting for the factorization's next row tile -- with the whole update in front of the counter a
		// row tile cost 3.3 us here against the 2.7 us at which they are emitted, and the lag reached 7 us by the last one
		{
			const int b = wave & 7;
			double xb[4];
#pragma unroll
			for(int kk = 0; kk < 4; ++ kk)
				xb[kk] = T[16 * J + (4 * kk + l4) + (16 * b + l15) * TS];
			auto row_update = [&](const int I) {
				const double *Ra = Rr + 16 * (I - J - 1) * PT;
				v4f64 d = (v4f64){0, 0, 0, 0};
#pragma unroll
				for(int kk = 0; kk < 4; ++ kk)
					d = __builtin_amdgcn_mfma_f64_16x16x4f64(Ra[(4 * kk + l4) + l15 * PT], xb[kk], d, 0, 0, 0);
				double *D = T + 16 * I + (16 * b) * TS;
#pragma unroll
				for(int r = 0; r < 4; ++ r)
					D[(l4 + 4 * r) + l15 * TS] -= d[r];
			};
			{
				const int I = J + 1 + (wave >> 3);
				if(I < 8)
					row_update(I);
			}
			// the row tile is out
			if(wave < 8)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			lds_barrier();
			if(tid == 0)
				__hip_atomic_store(a.pub + ti * a.Tc + tj, tag | (J + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for(int I = J + 3 + (wave >> 3); I < 8; I += 2)
				row_update(I);
			lds_barrier(); // (the staging of the next row tile overwrites Rr)
		}
		if(tj == ti + 1 && J == 7) stamp(5);
	}
}

} // namespace spp
