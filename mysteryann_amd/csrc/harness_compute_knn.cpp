// harness_compute_knn.cpp (binary: compute_knn) -- makes the two exact-kNN inputs of the reference's
// pipeline from .fbin files with the f32 MFMA brute-force generator (roar_exact_knn):
//   ground truth   u32 n; u32 K; ids[n*K]; f32 dists[n*K]     (read by load_gt_meta /
//                  load_gt_data_with_dist, include/efanna2e/util.h:84-155)          --with_dists
//   learn-base kNN u32 n; u32 K; ids[n*K]                      (read by LoadLearnBaseKNN,
//                  src/index_bipartite.cpp:2622-2639)
// The reference got these from the DiskANN submodule's compute_groundtruth tool (thirdparty/DiskANN is
// empty in the tree).  Distances in the file follow that tool's convention: squared L2, or the plain
// inner product for ip / cosine (tests/test_search_roargraph.cpp:47-50 negates the search's distances
// before comparing).
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "roargraph_amd/efanna2e.h"

int main(int argc, char **argv) {
    std::string base_file, query_file, out_file, dist;
    uint32_t K = 100;
    int device = 0;
    bool with_dists = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) {
                std::cerr << "the required argument for option '" << a << "' is missing\n";
                exit(-1);
            }
            return argv[++i];
        };
        if (a == "--base_data_path") base_file = next();
        else if (a == "--query_path") query_file = next();
        else if (a == "--out_path") out_file = next();
        else if (a == "--dist") dist = next();
        else if (a == "--K") K = (uint32_t)std::stoul(next());
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--with_dists") with_dists = true;
        else {
            std::cerr << "usage: compute_knn --dist <l2/ip/cosine> --base_data_path <fbin> --query_path <fbin> --out_path <ibin> "
                         "[--K 100] [--with_dists] [--device 0]\n";
            return a == "--help" || a == "-h" ? 0 : -1;
        }
    }
    if (base_file.empty() || query_file.empty() || out_file.empty() || dist.empty()) {
        std::cerr << "missing required option\n";
        return -1;
    }
    try {
        efanna2e::Metric metric = dist == "l2" ? efanna2e::L2 : dist == "cosine" ? efanna2e::COSINE : efanna2e::INNER_PRODUCT;
        uint32_t nb, db, nq, dq;
        efanna2e::load_meta<float>(base_file.c_str(), nb, db);
        efanna2e::load_meta<float>(query_file.c_str(), nq, dq);
        if (db != dq) throw std::runtime_error("base and query dimension mismatch");
        float *q = nullptr;
        efanna2e::load_data<float>(query_file.c_str(), nq, dq, q);
        const uint32_t qd = (dq + 7) / 8 * 8;
        efanna2e::IndexBipartite index(qd, nb, metric, nullptr, device);
        index.LoadSearchNeededData(base_file.c_str(), "");
        if (index.need_normalize)
            for (uint32_t i = 0; i < nq; ++i) efanna2e::normalize<float>(q + (size_t)i * qd, dq);
        std::vector<uint32_t> ids((size_t)nq * K);
        std::vector<float> dists((size_t)nq * K);
        index.ExactKnn(q, nq, K, ids.data(), dists.data());
        if (metric != efanna2e::L2)
            for (auto &v : dists) v = -v;
        std::ofstream out(out_file, std::ios::binary);
        if (!out.is_open()) throw std::runtime_error("cannot open file");
        out.write((char *)&nq, 4);
        out.write((char *)&K, 4);
        out.write((char *)ids.data(), (std::streamsize)ids.size() * 4);
        if (with_dists) out.write((char *)dists.data(), (std::streamsize)dists.size() * 4);
        std::cout << "wrote " << out_file << ": " << nq << " x " << K << (with_dists ? " ids + distances" : " ids") << std::endl;
        efanna2e::host_free(q);
    } catch (const std::exception &ex) {
        std::cerr << ex.what() << '\n';
        return -1;
    }
    return 0;
}
