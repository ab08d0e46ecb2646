// harness_build_roargraph.cpp (binary: test_build_roargraph) -- counterpart of the reference's
// tests/test_build_roargraph.cpp (own code; same flags :33-58, same sequence :83-137: load base +
// training queries, LoadLearnBaseKNN, BuildRoarGraph, SaveProjectionGraph), written against
// include/roargraph_amd/efanna2e.h.  boost::program_options is not in this image: flags are parsed by hand.
// With --bipartite_index_save_path it is the counterpart of tests/test_only_build_bipartite.cpp (:100-135):
// LoadLearnBaseKNN, BuildBipartite, Save.
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "roargraph_amd/efanna2e.h"

int main(int argc, char **argv) {
    std::string base_data_file, sampled_query_data_file, projection_index_save_file, learn_base_nn_file, dist,
        data_type = "float", bipartite_index_save_file;
    uint32_t M_sq = 32, M_pjbp = 32, L_pjpq = 32, num_threads = 1;  // defaults of :45-50
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) {
                std::cerr << "the required argument for option '" << a << "' is missing\n";
                exit(-1);
            }
            return argv[++i];
        };
        if (a == "--help" || a == "-h") {
            std::cout << "--data_type float --dist <l2/ip/cosine> --base_data_path <fbin> --sampled_query_data_path <fbin>\n"
                         "--projection_index_save_path <index> --learn_base_nn_path <ibin> [--M_sq 32] [--M_pjbp 32]\n"
                         "[--L_pjpq 32] [-T n] [--device 0]\n"
                         "[--bipartite_index_save_path <index>]  build and save the bipartite graph instead\n";
            return 0;
        } else if (a == "--data_type") data_type = next();
        else if (a == "--dist") dist = next();
        else if (a == "--base_data_path") base_data_file = next();
        else if (a == "--sampled_query_data_path") sampled_query_data_file = next();
        else if (a == "--projection_index_save_path") projection_index_save_file = next();
        else if (a == "--bipartite_index_save_path") bipartite_index_save_file = next();
        else if (a == "--learn_base_nn_path") learn_base_nn_file = next();
        else if (a == "--M_sq") M_sq = (uint32_t)std::stoul(next());
        else if (a == "--M_pjbp") M_pjbp = (uint32_t)std::stoul(next());
        else if (a == "--L_pjpq") L_pjpq = (uint32_t)std::stoul(next());
        else if (a == "-T" || a == "--num_threads") num_threads = (uint32_t)std::stoul(next());
        else if (a == "--device") device = std::stoi(next());
        else {
            std::cerr << "unrecognised option '" << a << "'\n";
            return -1;
        }
    }
    const bool bipartite = !bipartite_index_save_file.empty();
    if (dist.empty() || base_data_file.empty() || sampled_query_data_file.empty() ||
        (projection_index_save_file.empty() && !bipartite) || learn_base_nn_file.empty()) {
        std::cerr << "missing required option\n";
        return -1;
    }
    try {
        std::cout << "sampled query: " << sampled_query_data_file << std::endl;
        uint32_t base_num, base_dim, sq_num, sq_dim;
        efanna2e::load_meta<float>(base_data_file.c_str(), base_num, base_dim);
        efanna2e::load_meta<float>(sampled_query_data_file.c_str(), sq_num, sq_dim);
        efanna2e::Metric dist_metric = efanna2e::INNER_PRODUCT;
        if (dist == "l2") dist_metric = efanna2e::L2;
        else if (dist == "cosine") dist_metric = efanna2e::COSINE;
        else if (dist != "ip") {
            std::cout << "Unknown distance type: " << dist << std::endl;
            return -1;
        }
        float *data_bp = nullptr, *data_sq = nullptr;
        efanna2e::Parameters parameters;
        efanna2e::load_data<float>(base_data_file.c_str(), base_num, base_dim, data_bp);
        efanna2e::load_data<float>(sampled_query_data_file.c_str(), sq_num, sq_dim, data_sq);
        std::cout << "Index save path: " << projection_index_save_file << std::endl;
        efanna2e::IndexBipartite index_bipartite(base_dim, base_num + sq_num, dist_metric, nullptr, device);
        parameters.Set<uint32_t>("M_sq", M_sq);
        parameters.Set<uint32_t>("M_pjbp", M_pjbp);
        parameters.Set<uint32_t>("L_pjpq", L_pjpq);
        parameters.Set<uint32_t>("num_threads", num_threads);
        index_bipartite.LoadLearnBaseKNN(learn_base_nn_file.c_str());
        auto s = std::chrono::high_resolution_clock::now();
        if (bipartite) {
            index_bipartite.BuildBipartite(sq_num, data_sq, base_num, data_bp, parameters);
            std::chrono::duration<double> diff = std::chrono::high_resolution_clock::now() - s;
            std::cout << "indexing time: " << diff.count() << "\n";
            index_bipartite.Save(bipartite_index_save_file.c_str());
            std::cout << "Save index to " << bipartite_index_save_file << std::endl;
            std::cout << "Bipartite nodes: " << index_bipartite.Info("bipartite_nodes")
                      << " edges: " << index_bipartite.Info("bipartite_edges") << std::endl;
            efanna2e::host_free(data_bp);
            efanna2e::host_free(data_sq);
            return 0;
        }
        index_bipartite.BuildRoarGraph(sq_num, data_sq, base_num, data_bp, parameters);
        std::chrono::duration<double> diff = std::chrono::high_resolution_clock::now() - s;
        std::cout << "indexing time: " << diff.count() << "\n";
        index_bipartite.SaveProjectionGraph(projection_index_save_file.c_str());
        std::cout << "Save index to " << projection_index_save_file << std::endl;
        std::cout << "Projection degree avg: " << (double)index_bipartite.Info("nedges") / (double)base_num << std::endl;
        std::cout << "Projection degree max: " << index_bipartite.Info("max_degree") << std::endl;
        efanna2e::host_free(data_bp);
        efanna2e::host_free(data_sq);
    } catch (const std::exception &ex) {
        std::cerr << ex.what() << '\n';
        return -1;
    }
    return 0;
}
