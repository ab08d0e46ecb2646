// Compile test for the forwarding headers (include/compat): the quoted includes below are spelled as the reference's search harness
// spells them (tests/test_search_roargraph.cpp:14-18; its gtest / boost includes are not on the path), and every facade item that
// harness touches is used once with the reference's signature (:118-209).  Built by tests/test_capi_symbols.py with nothing but
// -I include/compat; run without arguments it touches no GPU.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "efanna2e/distance.h"
#include "efanna2e/neighbor.h"
#include "efanna2e/parameters.h"
#include "efanna2e/util.h"
#include "index_bipartite.h"

int main(int argc, char **argv) {
    efanna2e::Parameters parameters;
    parameters.Set<uint32_t>("num_threads", 4u);
    parameters.Set<uint32_t>("L_pq", 50u);
    if (parameters.Get<uint32_t>("L_pq") != 50u) return 1;
    const efanna2e::Metric metrics[] = {efanna2e::L2, efanna2e::INNER_PRODUCT, efanna2e::COSINE};
    if (argc < 5) {  // (no files given: the types and signatures have compiled, which is the test)
        std::cout << "compat headers ok: " << sizeof(metrics) / sizeof(metrics[0]) << " metrics" << std::endl;
        return 0;
    }
    const char *base_file = argv[1], *query_file = argv[2], *gt_file = argv[3], *index_file = argv[4];
    uint32_t base_num = 0, base_dim = 0, q_pts = 0, q_dim = 0, gt_pts = 0, gt_dim = 0;
    efanna2e::load_meta<float>(base_file, base_num, base_dim);
    efanna2e::load_meta<float>(query_file, q_pts, q_dim);
    float *query_data = nullptr;
    efanna2e::load_data<float>(query_file, q_pts, q_dim, query_data);
    float *aligned = efanna2e::data_align(query_data, q_pts, q_dim);
    uint32_t *gt_ids = nullptr;
    float *gt_dists = nullptr;
    efanna2e::load_gt_meta<uint32_t>(gt_file, gt_pts, gt_dim);
    efanna2e::load_gt_data_with_dist<uint32_t, float>(gt_file, gt_pts, gt_dim, gt_ids, gt_dists);
    efanna2e::IndexBipartite index(q_dim, base_num, metrics[1], nullptr);
    index.LoadSearchNeededData(base_file, "");
    index.LoadProjectionGraph(index_file);
    if (index.need_normalize)
        for (uint32_t i = 0; i < q_pts; ++i) efanna2e::normalize<float>(aligned + (size_t)i * q_dim, q_dim);
    index.InitVisitedListPool(4);
    const size_t k = 10;
    std::vector<uint32_t> res((size_t)q_pts * k);
    std::vector<std::vector<float>> res_dists(q_pts, std::vector<float>(k, 0.0f));
    uint64_t cmps = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : cmps)
    for (size_t i = 0; i < q_pts; ++i) {
        size_t qid = i;
        const std::pair<uint32_t, uint32_t> r = index.SearchRoarGraph(aligned + i * q_dim, k, qid, parameters, res.data() + i * k, res_dists[i]);
        cmps += r.first;
    }
    std::cout << "searched " << q_pts << " queries, " << cmps << " distance evaluations" << std::endl;
    delete[] aligned;
    delete[] gt_ids;
    delete[] gt_dists;
    return 0;
}
