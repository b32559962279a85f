// host-only check of the orderings on a ring-band block graph (no GPU needed): permutation validity,
// elimination-tree height and block-level fill of minimum degree vs nested dissection
#include "../slam_plus_plus_amd/csrc/spp_internal.h"
#include <stdio.h>
#include <set>
using namespace spp;

static void stats(int64_t nb, const std::vector<std::vector<int32_t> > &adj, const std::vector<int64_t> &order, const char *name)
{
	std::vector<int64_t> inv(nb, -1);
	for(int64_t k = 0; k < nb; ++ k) {
		if(order[k] < 0 || order[k] >= nb || inv[order[k]] >= 0) { printf("%s: NOT a permutation at %lld\n", name, (long long)k); return; }
		inv[order[k]] = k;
	}
	// symbolic elimination with sets (small graphs only)
	std::vector<std::set<int32_t> > rs(nb);
	for(int64_t v = 0; v < nb; ++ v)
		for(size_t q = 0; q < adj[v].size(); ++ q) {
			int64_t a = inv[v], b = inv[adj[v][q]];
			if(a < b) rs[a].insert((int32_t)b);
		}
	double nnz = 0, flops = 0;
	std::vector<int32_t> depth(nb, 1);
	int64_t height = 0;
	for(int64_t j = 0; j < nb; ++ j) {
		nnz += rs[j].size() + 1;
		flops += (double)(rs[j].size() + 1) * (rs[j].size() + 1);
		if(!rs[j].empty()) {
			int32_t p = *rs[j].begin();
			for(auto it = std::next(rs[j].begin()); it != rs[j].end(); ++ it) rs[p].insert(*it);
			depth[p] = std::max(depth[p], depth[j] + 1);
		}
		height = std::max<int64_t>(height, depth[j]);
	}
	printf("%s: nnzb(R) %.0f  block flops %.3g  etree height %lld\n", name, nnz, flops, (long long)height);
}

int main(int argc, char **argv)
{
	const int64_t nb = argc > 1 ? atoll(argv[1]) : 2000;
	const int half = argc > 2 ? atoi(argv[2]) : 10;
	const bool ring = argc > 3 ? atoi(argv[3]) != 0 : true;
	std::vector<std::vector<int32_t> > adj(nb);
	std::vector<int64_t> cp(nb + 1, 0), ri;
	for(int64_t j = 0; j < nb; ++ j) {
		std::set<int64_t> rows;
		for(int d = 1; d <= half; ++ d) {
			int64_t i = j - d;
			if(i < 0) { if(!ring) continue; i += nb; }
			int64_t a = std::min(i, j), b = std::max(i, j);
			if(b == j) rows.insert(a);
			else { /* belongs to column b */ }
		}
		if(ring)
			for(int d = 1; d <= half; ++ d) { int64_t i = (j + d) % nb; if(i < j) rows.insert(i); }
		for(auto r : rows) { ri.push_back(r); adj[r].push_back((int32_t)j); adj[j].push_back((int32_t)r); }
		ri.push_back(j);
		cp[j + 1] = (int64_t)ri.size();
	}
	std::vector<int64_t> amd, nd;
	min_degree_order(nb, cp.data(), ri.data(), amd);
	stats(nb, adj, amd, "amd");
	nested_dissection_order(nb, cp.data(), ri.data(), nd);
	stats(nb, adj, nd, "nd ");
	return 0;
}
namespace spp { void dense_reserve(spp_ctx *, int64_t) {} } // link stub: the plan builder is not exercised here
