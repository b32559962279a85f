/*
 * oracle/spp_oracle.c -- TEST INFRASTRUCTURE ONLY. See spp_oracle.h for the scope statement.
 * Plain C99, single thread, no dependencies. Each function names the reference lines it restates.
 */
#include "spp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- tiny dense helpers (column-major) --------------------------------------------------- */

/* C(m x n) = A^T(m x k)^T... : C = At^T * B where At is k x m, B is k x n */
static void mm_tn(int m, int n, int k, const double *At, const double *B, double *C)
{
	for(int j = 0; j < n; ++ j)
		for(int i = 0; i < m; ++ i) {
			double s = 0;
			for(int l = 0; l < k; ++ l)
				s += At[l + i * k] * B[l + j * k];
			C[i + j * m] = s;
		}
}

/* C(m x n) = A(m x k) * B(k x n) */
static void mm_nn(int m, int n, int k, const double *A, const double *B, double *C)
{
	for(int j = 0; j < n; ++ j)
		for(int i = 0; i < m; ++ i) {
			double s = 0;
			for(int l = 0; l < k; ++ l)
				s += A[i + l * m] * B[l + j * k];
			C[i + j * m] = s;
		}
}

/* ---- a-1: per-edge Hessian blocks ---------------------------------------------------------
 * BaseTypes_Binary.h:768-774  T = J0^T Omega
 *                  :776-808  H01 = T J1, or (J1^T T^T) when the vertex ids are reversed
 *                  :810-815  H00 = (T J0).selfadjointView<Upper>()  (upper half mirrored)
 *                  :819-823  g0 = T r
 *                  :825-846  H11 = (J1^T Omega J1).selfadjointView<Upper>(), g1 = J1^T (Omega r) */
/* Robust edges (b_is_robust_edge, BaseTypes_Binary.h:768-774, 821-846): one weight w per edge, the value of the edge's
 * f_RobustWeight(r). T = J0^T Omega w replaces J0^T Omega (so H01 and H00 carry w once), g0 = T r w -- w TWICE, as the
 * reference writes it --, H11 = J1^T Omega J1 w and g1 = J1^T (Omega r) w. w == NULL: plain edges. */
void orc_edge_hessians_w(int d0, int d1, int rd, int64_t ne,
	const double *J0, const double *J1, const double *Om, const double *r, const double *w,
	const uint8_t *reversed,
	double *H01, double *H00, double *H11, double *g0, double *g1)
{
	double T[6 * 6], T1[6 * 6], tmp[6 * 6], Or[6];
	for(int64_t e = 0; e < ne; ++ e) {
		const double *j0 = J0 + e * rd * d0, *j1 = J1 + e * rd * d1;
		const double *om = Om + e * rd * rd, *re = r + e * rd;
		mm_tn(d0, rd, rd, j0, om, T); /* T = J0^T Om  (d0 x rd) */
		if(w)
			for(int i = 0; i < d0 * rd; ++ i)
				T[i] *= w[e]; /* :771 t_H0_sigma_inv = J0^T * Sigma^-1 * w */
		double *h01 = H01 + e * d0 * d1;
		if(reversed && reversed[e]) {
			/* t_HtSiH (d1 x d0) = J1^T * T^T */
			for(int j = 0; j < d0; ++ j)
				for(int i = 0; i < d1; ++ i) {
					double s = 0;
					for(int l = 0; l < rd; ++ l)
						s += j1[l + i * rd] * T[j + l * d0];
					h01[i + j * d1] = s;
				}
		} else
			mm_nn(d0, d1, rd, T, j1, h01);
		mm_nn(d0, d0, rd, T, j0, tmp);
		double *h00 = H00 + e * d0 * d0;
		for(int j = 0; j < d0; ++ j)
			for(int i = 0; i < d0; ++ i)
				h00[i + j * d0] = (i <= j)? tmp[i + j * d0] : tmp[j + i * d0];
		mm_nn(d0, 1, rd, T, re, g0 + e * d0);
		if(w)
			for(int i = 0; i < d0; ++ i)
				g0[e * d0 + i] *= w[e]; /* :821 t_H0_sigma_inv * v_error * w */
		/* J1^T Om J1: Eigen evaluates (J1^T * Om) * J1 left to right */
		mm_tn(d1, rd, rd, j1, om, T1);
		mm_nn(d1, d1, rd, T1, j1, tmp);
		if(w)
			for(int i = 0; i < d1 * d1; ++ i)
				tmp[i] *= w[e]; /* :829, :835 J1^T Sigma^-1 J1 * w */
		double *h11 = H11 + e * d1 * d1;
		for(int j = 0; j < d1; ++ j)
			for(int i = 0; i < d1; ++ i)
				h11[i + j * d1] = (i <= j)? tmp[i + j * d1] : tmp[j + i * d1];
		mm_nn(rd, 1, rd, om, re, Or);
		mm_tn(d1, 1, rd, j1, Or, g1 + e * d1);
		if(w)
			for(int i = 0; i < d1; ++ i)
				g1[e * d1 + i] *= w[e]; /* :844 J1^T (Sigma^-1 r) * w */
	}
}

void orc_edge_hessians(int d0, int d1, int rd, int64_t ne,
	const double *J0, const double *J1, const double *Om, const double *r,
	const uint8_t *reversed,
	double *H01, double *H00, double *H11, double *g0, double *g1)
{
	orc_edge_hessians_w(d0, d1, rd, ne, J0, J1, Om, r, 0, reversed, H01, H00, H11, g0, g1);
}

/* ---- a-2: reduction plan -------------------------------------------------------------------
 * NonlinearSolver_Lambda_Base.h:598-604: the first source is assigned, the others are added in
 * list order (= edge order), which makes the sums bit-reproducible. */
void orc_reduce(int64_t n_dst, const int64_t *dst_off, const int32_t *dst_len,
	const int64_t *list_ptr, const int64_t *src_off, const double *src, double *dst)
{
	for(int64_t d = 0; d < n_dst; ++ d) {
		double *o = dst + dst_off[d];
		int len = dst_len[d];
		int64_t b = list_ptr[d], e = list_ptr[d + 1];
		if(b == e)
			continue; /* nothing reduces into this block: leave as is */
		const double *s = src + src_off[b];
		for(int i = 0; i < len; ++ i)
			o[i] = s[i];
		for(int64_t p = b + 1; p < e; ++ p) {
			s = src + src_off[p];
			for(int i = 0; i < len; ++ i)
				o[i] += s[i];
		}
	}
}

/* ---- a-5: numeric part of the symmetric permutation ----------------------------------------
 * BlockMatrix.cpp:8267-8281: a block that would land below the diagonal is stored transposed. */
void orc_permute_values(int64_t nnzb, const int64_t *src_off, const int64_t *dst_off,
	const int32_t *rows, const int32_t *cols, const uint8_t *transpose,
	const double *src, double *dst)
{
	for(int64_t p = 0; p < nnzb; ++ p) {
		const double *s = src + src_off[p];
		double *d = dst + dst_off[p];
		int m = rows[p], n = cols[p];
		if(!transpose[p])
			memcpy(d, s, sizeof(double) * m * n);
		else {
			for(int j = 0; j < n; ++ j)
				for(int i = 0; i < m; ++ i)
					d[j + i * n] = s[i + j * m];
		}
	}
}

/* ---- a-6: elimination tree (Liu, with path compression through "highest dependence") -------
 * BlockMatrix.cpp:9403-9451 */
void orc_etree(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, int64_t *parent)
{
	int64_t *anc = (int64_t*)malloc(sizeof(int64_t) * (nb? nb : 1));
	for(int64_t j = 0; j < nb; ++ j) {
		parent[j] = -1;
		anc[j] = -1;
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
			int64_t i = row_idx[p];
			if(i >= j)
				break;
			do {
				int64_t next = anc[i];
				anc[i] = j;
				if(next == -1) {
					parent[i] = j;
					break;
				}
				i = next;
			} while(i < j);
		}
	}
	free(anc);
}

/* ereach of column j (BlockMatrix.cpp:9453-9545): pattern of R(0:j-1, j), topologically ordered
 * at the END of stack[0..nb); returns the index of the first entry */
static int64_t ereach(int64_t nb, int64_t j, const int64_t *col_ptr, const int64_t *row_idx,
	const int64_t *parent, int64_t *stack, uint8_t *mark)
{
	int64_t first = nb;
	mark[j] = 1;
	for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
		int64_t i = row_idx[p];
		if(i > j)
			break;
		int64_t len = 0;
		for(; !mark[i]; i = parent[i]) {
			stack[len ++] = i;
			mark[i] = 1;
		}
		while(len)
			stack[-- first] = stack[-- len];
	}
	for(int64_t u = first; u < nb; ++ u)
		mark[stack[u]] = 0;
	mark[j] = 0;
	return first;
}

static int cmp_i64(const void *a, const void *b)
{
	int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
	return (x > y) - (x < y);
}

int64_t orc_chol_symbolic(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx,
	const int64_t *parent, int64_t *r_col_ptr, int64_t *r_row_idx)
{
	int64_t *stack = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	uint8_t *mark = (uint8_t*)calloc(nb + 1, 1);
	int64_t nnzb = 0;
	for(int64_t j = 0; j < nb; ++ j) {
		int64_t first = ereach(nb, j, col_ptr, row_idx, parent, stack, mark);
		r_col_ptr[j] = nnzb;
		if(r_row_idx) {
			int64_t cnt = nb - first;
			memcpy(r_row_idx + nnzb, stack + first, sizeof(int64_t) * cnt);
			qsort(r_row_idx + nnzb, cnt, sizeof(int64_t), cmp_i64);
			r_row_idx[nnzb + cnt] = j;
		}
		nnzb += nb - first + 1;
	}
	r_col_ptr[nb] = nnzb;
	free(stack);
	free(mark);
	return nnzb;
}

/* unblocked upper Cholesky of a small d x d block (only the upper half is read), R^T R = A.
 * Stands in for Eigen::LLT<MatrixXd, Upper> on the block (BlockMatrix.cpp:9762-9773): same
 * pivot test (non-positive pivot = failure), strictly lower part of the result zeroed the way
 * chol.matrixU() yields it. */
static int chol_upper_small(int d, double *A)
{
	for(int j = 0; j < d; ++ j) {
		double s = A[j + j * d];
		for(int k = 0; k < j; ++ k)
			s -= A[k + j * d] * A[k + j * d];
		if(!(s > 0))
			return 1;
		double rjj = sqrt(s);
		A[j + j * d] = rjj;
		for(int c = j + 1; c < d; ++ c) {
			double t = A[j + c * d];
			for(int k = 0; k < j; ++ k)
				t -= A[k + j * d] * A[k + c * d];
			A[j + c * d] = t / rjj;
		}
	}
	for(int j = 0; j < d; ++ j)
		for(int i = j + 1; i < d; ++ i)
			A[i + j * d] = 0;
	return 0;
}

/* ---- a-7: up-looking block Cholesky ------------------------------------------------------------
 * BlockMatrix.cpp:9595-9729: for k in ereach(j): R_kj = A_kj - sum_{i<k} R_ik^T R_ij (two-pointer
 * merge of the sorted block lists of columns k and j), then R_kk^T X = R_kj by forward
 * substitution (triangularView<Upper>().transpose().solveInPlace, :9726).
 * :9732-9773: R_jj = chol_upper(A_jj - sum_i R_ij^T R_ij) (upper half only, sorted row order). */
int orc_cholesky(int64_t nb, const int32_t *dim,
	const int64_t *a_col_ptr, const int64_t *a_row_idx, const int64_t *a_blk_off, const double *a_vals,
	const int64_t *parent,
	const int64_t *r_col_ptr, const int64_t *r_row_idx, const int64_t *r_blk_off, double *r_vals)
{
	int64_t *stack = (int64_t*)malloc(sizeof(int64_t) * (nb + 1));
	uint8_t *mark = (uint8_t*)calloc(nb + 1, 1);
	uint8_t *done = (uint8_t*)calloc(nb + 1, 1); /* which blocks of column j are already final */
	int ret = 0;
	for(int64_t j = 0; j < nb && !ret; ++ j) {
		const int dj = dim[j];
		int64_t first = ereach(nb, j, a_col_ptr, a_row_idx, parent, stack, mark);
		const int64_t rb = r_col_ptr[j], re = r_col_ptr[j + 1]; /* last one is the diagonal */
		for(int64_t u = first; u < nb; ++ u) {
			const int64_t k = stack[u];
			const int dk = dim[k];
			/* locate (k, j) in R's column j */
			int64_t lo = rb, hi = re - 1;
			while(lo < hi) {
				int64_t mid = (lo + hi) / 2;
				if(r_row_idx[mid] < k) lo = mid + 1; else hi = mid;
			}
			const int64_t pkj = lo;
			double *Rkj = r_vals + r_blk_off[pkj];
			/* A(k, j) or zero */
			{
				int64_t alo = a_col_ptr[j], ahi = a_col_ptr[j + 1];
				while(alo < ahi) {
					int64_t mid = (alo + ahi) / 2;
					if(a_row_idx[mid] < k) alo = mid + 1; else ahi = mid;
				}
				if(alo < a_col_ptr[j + 1] && a_row_idx[alo] == k)
					memcpy(Rkj, a_vals + a_blk_off[alo], sizeof(double) * dk * dj);
				else
					memset(Rkj, 0, sizeof(double) * dk * dj);
			}
			/* sparse dot of columns k and j over rows i < k (only blocks of j already computed) */
			{
				int64_t pk = r_col_ptr[k];
				const int64_t pk_end = r_col_ptr[k + 1] - 1; /* exclude diagonal of k */
				for(int64_t pj = rb; pj < pkj; ++ pj) {
					if(!done[pj - rb])
						continue; /* not produced yet: cannot happen for i in ereach before k */
					const int64_t i = r_row_idx[pj];
					while(pk < pk_end && r_row_idx[pk] < i)
						++ pk;
					if(pk < pk_end && r_row_idx[pk] == i) {
						const int di = dim[i];
						const double *Rik = r_vals + r_blk_off[pk], *Rij = r_vals + r_blk_off[pj];
						for(int c = 0; c < dj; ++ c)
							for(int a = 0; a < dk; ++ a) {
								double s = 0;
								for(int l = 0; l < di; ++ l)
									s += Rik[l + a * di] * Rij[l + c * di];
								Rkj[a + c * dk] -= s;
							}
					}
				}
			}
			/* R_kk^T X = R_kj */
			{
				const double *Rkk = r_vals + r_blk_off[r_col_ptr[k + 1] - 1];
				for(int c = 0; c < dj; ++ c)
					for(int a = 0; a < dk; ++ a) {
						double s = Rkj[a + c * dk];
						for(int l = 0; l < a; ++ l)
							s -= Rkk[l + a * dk] * Rkj[l + c * dk];
						Rkj[a + c * dk] = s / Rkk[a + a * dk];
					}
			}
			done[pkj - rb] = 1;
		}
		/* diagonal block */
		{
			double *Rjj = r_vals + r_blk_off[re - 1];
			int64_t alo = a_col_ptr[j + 1] - 1; /* diagonal is the last block of A's column */
			memcpy(Rjj, a_vals + a_blk_off[alo], sizeof(double) * dj * dj);
			for(int64_t pj = rb; pj < re - 1; ++ pj) {
				const int di = dim[r_row_idx[pj]];
				const double *Rij = r_vals + r_blk_off[pj];
				for(int c = 0; c < dj; ++ c)
					for(int a = 0; a <= c; ++ a) {
						double s = 0;
						for(int l = 0; l < di; ++ l)
							s += Rij[l + a * di] * Rij[l + c * di];
						Rjj[a + c * dj] -= s;
					}
			}
			if(chol_upper_small(dj, Rjj))
				ret = 1;
		}
		memset(done, 0, re - rb);
	}
	free(stack);
	free(mark);
	free(done);
	return ret;
}

/* ---- a-8: triangular solves ------------------------------------------------------------------ */
int orc_utsolve(int64_t nb, const int32_t *dim, const int64_t *base,
	const int64_t *col_ptr, const int64_t *row_idx, const int64_t *blk_off, const double *vals, double *x)
{
	for(int64_t j = 0; j < nb; ++ j) { /* BlockMatrix.cpp:8641-8716 */
		const int dj = dim[j];
		double *xj = x + base[j];
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1] - 1; ++ p) {
			const int64_t i = row_idx[p];
			const int di = dim[i];
			const double *B = vals + blk_off[p], *xi = x + base[i];
			for(int c = 0; c < dj; ++ c) {
				double s = 0;
				for(int l = 0; l < di; ++ l)
					s += xi[l] * B[l + c * di];
				xj[c] -= s;
			}
		}
		const double *D = vals + blk_off[col_ptr[j + 1] - 1];
		for(int i = 0; i < dj; ++ i) {
			double f = xj[i];
			for(int l = 0; l < i; ++ l)
				f -= D[l + i * dj] * xj[l];
			if(D[i + i * dj] == 0)
				return 1;
			xj[i] = f / D[i + i * dj];
		}
	}
	return 0;
}

int orc_usolve(int64_t nb, const int32_t *dim, const int64_t *base,
	const int64_t *col_ptr, const int64_t *row_idx, const int64_t *blk_off, const double *vals, double *x)
{
	for(int64_t j = nb; j > 0;) { /* BlockMatrix.cpp:8991-9060 */
		-- j;
		const int dj = dim[j];
		double *xj = x + base[j];
		const double *D = vals + blk_off[col_ptr[j + 1] - 1];
		for(int i = dj; i > 0;) {
			-- i;
			if(D[i + i * dj] == 0)
				return 1;
			double f = (xj[i] /= D[i + i * dj]);
			for(int l = 0; l < i; ++ l)
				xj[l] -= D[l + i * dj] * f;
		}
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1] - 1; ++ p) {
			const int64_t i = row_idx[p];
			const int di = dim[i];
			const double *B = vals + blk_off[p];
			double *xi = x + base[i];
			for(int l = 0; l < di; ++ l) {
				double s = 0;
				for(int c = 0; c < dj; ++ c)
					s += B[l + c * di] * xj[c];
				xi[l] -= s;
			}
		}
	}
	return 0;
}

/* ---- a-18: dense LLT (upper) + solve ------------------------------------------------------------
 * LinearSolver_Schur.cpp:2317-2327: Eigen::LLT<MatrixXd, Upper> then .solve(); only the upper
 * triangle is referenced. Left-looking by columns (dot-product form), exact pivots test. */
int orc_dense_llt_solve(int64_t n, double *A, int64_t ld, double *b)
{
	for(int64_t j = 0; j < n; ++ j) {
		double *cj = A + j * ld;
		for(int64_t i = 0; i < j; ++ i) {
			const double *ci = A + i * ld;
			double s = cj[i];
			for(int64_t k = 0; k < i; ++ k)
				s -= ci[k] * cj[k];
			cj[i] = s / ci[i];
		}
		double s = cj[j];
		for(int64_t k = 0; k < j; ++ k)
			s -= cj[k] * cj[k];
		if(!(s > 0))
			return 1;
		cj[j] = sqrt(s);
	}
	for(int64_t j = 0; j < n; ++ j) { /* R^T y = b */
		const double *cj = A + j * ld;
		double s = b[j];
		for(int64_t k = 0; k < j; ++ k)
			s -= cj[k] * b[k];
		b[j] = s / cj[j];
	}
	for(int64_t j = n; j > 0;) { /* R x = y */
		-- j;
		const double *cj = A + j * ld;
		double f = (b[j] /= cj[j]);
		for(int64_t k = 0; k < j; ++ k)
			b[k] -= cj[k] * f;
	}
	return 0;
}

/* ---- a-15: 3x3 inverse, cofactor formula (what Eigen's fixed-size .inverse() does for 3x3) ---- */
void orc_inverse3(const double *m, double *o)
{
	/* column-major: M(i,j) = m[i + 3 j]; cof(i,j) = M(i1,j1) M(i2,j2) - M(i1,j2) M(i2,j1) with
	 * i1 = (i+1)%3, i2 = (i+2)%3 (same for j): the cyclic form needs no sign factor.
	 * det is expanded along column 0, result(i,j) = cof(j,i) / det with ONE reciprocal. */
#define M_(i, j) m[(i) + 3 * (j)]
#define COF_(i, j) (M_(((i) + 1) % 3, ((j) + 1) % 3) * M_(((i) + 2) % 3, ((j) + 2) % 3) - \
	M_(((i) + 1) % 3, ((j) + 2) % 3) * M_(((i) + 2) % 3, ((j) + 1) % 3))
	const double c00 = COF_(0, 0), c10 = COF_(1, 0), c20 = COF_(2, 0);
	const double det = c00 * M_(0, 0) + c10 * M_(1, 0) + c20 * M_(2, 0);
	const double id = 1.0 / det;
	o[0 + 3 * 0] = c00 * id;
	o[0 + 3 * 1] = c10 * id;
	o[0 + 3 * 2] = c20 * id;
	o[1 + 3 * 0] = COF_(0, 1) * id;
	o[1 + 3 * 1] = COF_(1, 1) * id;
	o[1 + 3 * 2] = COF_(2, 1) * id;
	o[2 + 3 * 0] = COF_(0, 2) * id;
	o[2 + 3 * 1] = COF_(1, 2) * id;
	o[2 + 3 * 2] = COF_(2, 2) * id;
#undef COF_
#undef M_
}

/* ---- a-13 .. a-18: Schur complement solve -------------------------------------------------------
 * LinearSolver_Schur.h:1699-1709 slice into A (poses), U (pose x landmark), C (landmark diagonal)
 *   :1721-1735  Cinv = -(C^-1) block by block
 *   :1737-1745  W = U * Cinv               (= -U C^-1, structure of U)
 *   :1757-1767  S = W * U^T (upper only) + A
 *   :1811-1830  (x | l) = eta ; x += W l
 *   :1839-1853  dense LLT solve S dx = x
 *   :1867-1881  l = -l ; l += U^T dx ; dl = Cinv l
 * Lambda is expected already in guided order (LinearSolver_Schur.cpp:771-838): poses first. */
int orc_schur_solve(int64_t nb, int64_t nc, const int32_t *dim, const int64_t *base,
	const int64_t *col_ptr, const int64_t *row_idx, const int64_t *blk_off, const double *vals,
	double *rhs, double *S_out)
{
	const int64_t np_ = base[nc], n = base[nb];
	int dl = (nb > nc)? dim[nc] : 0;
	if(dl != 3)
		return -1;
	for(int64_t j = nc; j < nb; ++ j) {
		if(dim[j] != 3)
			return -1;
		if(row_idx[col_ptr[j + 1] - 1] != j)
			return -1;
		if(col_ptr[j + 1] - col_ptr[j] > 1 && row_idx[col_ptr[j + 1] - 2] >= nc)
			return -1; /* C must be block diagonal */
	}
	double *S = (double*)calloc((size_t)np_ * np_, sizeof(double));
	if(!S)
		return -2;
	/* A -> S (upper blocks) : Convert_to_Dense leaves the lower triangle zero */
	for(int64_t j = 0; j < nc; ++ j)
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
			const int64_t i = row_idx[p];
			const int di = dim[i], dj = dim[j];
			const double *B = vals + blk_off[p];
			for(int c = 0; c < dj; ++ c)
				for(int a = 0; a < di; ++ a)
					S[(base[i] + a) + (base[j] + c) * np_] = B[a + c * di];
		}
	double *x = rhs, *l = rhs + np_;
	double *W = (double*)malloc(sizeof(double) * 6 * 3 * 64);
	int64_t wcap = 64;
	double *Cinv_all = (double*)malloc(sizeof(double) * 9 * (nb - nc + 1));
	for(int64_t j = nc; j < nb; ++ j) {
		const int64_t pb = col_ptr[j], pe = col_ptr[j + 1] - 1; /* U blocks of landmark j */
		double *Ci = Cinv_all + 9 * (j - nc);
		orc_inverse3(vals + blk_off[pe], Ci);
		for(int q = 0; q < 9; ++ q)
			Ci[q] = -Ci[q]; /* Scale(-1), :1735 */
		const int64_t k = pe - pb;
		if(k > wcap) {
			wcap = 2 * k;
			W = (double*)realloc(W, sizeof(double) * 6 * 3 * wcap);
		}
		/* W_i = U_ij * Cinv  (d_i x 3) */
		for(int64_t p = pb; p < pe; ++ p) {
			const int di = dim[row_idx[p]];
			mm_nn(di, 3, 3, vals + blk_off[p], Ci, W + 18 * (p - pb));
		}
		/* S(i1, i2) += W_i1 * U_i2^T for i1 <= i2 (upper only, b_upper_diag_only) */
		for(int64_t p2 = pb; p2 < pe; ++ p2) {
			const int64_t i2 = row_idx[p2];
			const int d2 = dim[i2];
			const double *U2 = vals + blk_off[p2];
			for(int64_t p1 = pb; p1 <= p2; ++ p1) {
				const int64_t i1 = row_idx[p1];
				const int d1 = dim[i1];
				const double *W1 = W + 18 * (p1 - pb);
				for(int c = 0; c < d2; ++ c)
					for(int a = 0; a < d1; ++ a) {
						double s = 0;
						for(int q = 0; q < 3; ++ q)
							s += W1[a + q * d1] * U2[c + q * d2];
						S[(base[i1] + a) + (base[i2] + c) * np_] += s;
					}
			}
		}
		/* x += W l_j */
		const double *lj = l + (base[j] - np_);
		for(int64_t p = pb; p < pe; ++ p) {
			const int64_t i = row_idx[p];
			const int di = dim[i];
			const double *Wi = W + 18 * (p - pb);
			for(int a = 0; a < di; ++ a) {
				double s = 0;
				for(int q = 0; q < 3; ++ q)
					s += Wi[a + q * di] * lj[q];
				x[base[i] + a] += s;
			}
		}
	}
	if(S_out)
		memcpy(S_out, S, sizeof(double) * np_ * np_);
	int ret = orc_dense_llt_solve(np_, S, np_, x);
	if(!ret) {
		for(int64_t j = nc; j < nb; ++ j) {
			const int64_t pb = col_ptr[j], pe = col_ptr[j + 1] - 1;
			double *lj = l + (base[j] - np_);
			double t[3] = {-lj[0], -lj[1], -lj[2]};
			for(int64_t p = pb; p < pe; ++ p) {
				const int64_t i = row_idx[p];
				const int di = dim[i];
				const double *Ui = vals + blk_off[p];
				for(int q = 0; q < 3; ++ q) {
					double s = 0;
					for(int a = 0; a < di; ++ a)
						s += x[base[i] + a] * Ui[a + q * di];
					t[q] += s;
				}
			}
			const double *Ci = Cinv_all + 9 * (j - nc);
			for(int a = 0; a < 3; ++ a)
				lj[a] = Ci[a] * t[0] + Ci[a + 3] * t[1] + Ci[a + 6] * t[2];
		}
	}
	(void)n;
	free(S);
	free(W);
	free(Cinv_all);
	return ret;
}
