// harness_search_roargraph.cpp (binary: test_search_roargraph) -- harness counterpart of the reference's
// tests/test_search_roargraph.cpp (own code; same flags, same load sequence :118-173, same
// per-L_pq loop :191-237, same stdout columns and CSV row `L_pq,qps,avg_cmps,mean_latency_ms,
// recall,avg_hops`), written against include/roargraph_amd/efanna2e.h.
//
// Differences, all forced by the device: the timed region is ONE batched launch over all queries
// (--per_query restores the reference's `omp parallel for` over single-query calls); QPS uses a
// microsecond clock (the reference truncates to whole milliseconds, :212-213, which is 0 here for
// small L_pq).  boost::program_options is not available in this image: flags are parsed by hand.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "roargraph_amd/efanna2e.h"

// tests/test_search_roargraph.cpp:23-36
static float ComputeRecall(uint32_t q_num, uint32_t k, uint32_t gt_dim, const uint32_t *res, const uint32_t *gt) {
    uint32_t total = 0;
    for (uint32_t i = 0; i < q_num; i++)
        for (uint32_t a = 0; a < k; a++) {
            uint32_t p = gt[(size_t)i * gt_dim + a];
            if (std::find(res + (size_t)i * k, res + (size_t)i * k + k, p) != res + (size_t)i * k + k) total++;
        }
    return (float)total / (float)(k * q_num);
}

// Relative distance error of the returned neighbours against the ground-truth distances; the
// reference defines it in tests/test_search_roargraph.cpp:38-62 (never called there) and prints it
// from the bipartite driver (tests/test_search_bipartite.cpp:226-231).  IP results are negated
// back to dot products, COSINE to 2 (1 - cos), as there.
static double ComputeRderr(const float *gt_dist, uint32_t gt_dim, const float *res_d, uint32_t q_num, uint32_t k,
                           efanna2e::Metric metric) {
    double total = 0;
    for (uint32_t i = 0; i < q_num; i++) {
        double err = 0.0;
        for (uint32_t j = 0; j < k; j++) {
            double r = res_d[(size_t)i * k + j];
            if (metric == efanna2e::INNER_PRODUCT) r = -r;
            else if (metric == efanna2e::COSINE) r = 2.0 * (1.0 - (-r));
            const double g = gt_dist[(size_t)i * gt_dim + j];
            err += std::fabs(r - g) / g;
        }
        total += err / k;
    }
    return total / q_num;
}

static void usage() {
    std::cout << "Arguments:\n"
                 "  --data_type float            (only float is supported, as in the reference's driver)\n"
                 "  --dist <l2/ip/cosine>\n"
                 "  --base_data_path <fbin>\n"
                 "  --query_path <fbin>\n"
                 "  --gt_path <ibin with distances>\n"
                 "  --projection_index_save_path <index>\n"
                 "  --L_pq <v1> [v2 ...]\n"
                 "  --k <k>                      (default 1)\n"
                 "  --evaluation_save_path <csv>\n"
                 "  -T / --num_threads <n>       (per_query mode only)\n"
                 "  --per_query                  reference loop shape: one SearchRoarGraph call per query\n"
                 "  --visited <auto|exact|lds>   visited-set policy (default auto)\n"
                 "  --rderr                      also print / save the relative distance error (7th CSV column)\n"
                 "  --bipartite_index_save_path <index>  search the bipartite graph instead (SearchBipartiteGraph)\n"
                 "  --entry_seed <n>             seed of the bipartite entry-point draws\n"
                 "  --device <ordinal>\n"
                 "  --gpus <g>                   index replicated on devices device .. device + g - 1, the query block split by\n"
                 "                               query over them (no collective); 0 = every visible device; default 1\n"
                 "  --devices <d0,d1,...>        explicit replica placement (a device may repeat); overrides --gpus\n"
                 "  --shards <d0,d1,...>         index-sharded configuration: --base_data_path and --projection_index_save_path are\n"
                 "                               comma separated lists, one file per shard (contiguous id ranges, local ids);\n"
                 "                               shard s lives on device ds; every query is searched on every shard, the per-shard\n"
                 "                               lists meet in one RCCL all-gather and are merged (ids reported are global)\n";
}

int main(int argc, char **argv) {
    // (launches that overlap on several HIP streams -- --gpus, --shards -- want a hardware queue each; read by the HIP runtime when it starts)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    std::string base_data_file, query_file, gt_file, projection_index_save_file, data_type = "float", dist;
    std::string evaluation_save_path, visited = "auto", bipartite_index_save_file;
    long long entry_seed = -1;
    bool want_rderr = false;
    std::vector<uint32_t> L_vec;
    uint32_t num_threads = omp_get_num_procs(), k = 1;
    int device = 0, gpus = 1;
    std::vector<int> device_list, shard_devices;
    bool per_query = false;
    auto split_ints = [](const std::string &v) {
        std::vector<int> out;
        for (size_t p0 = 0; p0 <= v.size();) {
            size_t p1 = v.find(',', p0);
            if (p1 == std::string::npos) p1 = v.size();
            out.push_back(std::stoi(v.substr(p0, p1 - p0)));
            p0 = p1 + 1;
        }
        return out;
    };
    auto split_names = [](const std::string &v) {
        std::vector<std::string> out;
        for (size_t p0 = 0; p0 <= v.size();) {
            size_t p1 = v.find(',', p0);
            if (p1 == std::string::npos) p1 = v.size();
            out.push_back(v.substr(p0, p1 - p0));
            p0 = p1 + 1;
        }
        return out;
    };
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
            usage();
            return 0;
        } else if (a == "--data_type") data_type = next();
        else if (a == "--dist") dist = next();
        else if (a == "--base_data_path") base_data_file = next();
        else if (a == "--query_path") query_file = next();
        else if (a == "--gt_path") gt_file = next();
        else if (a == "--projection_index_save_path") projection_index_save_file = next();
        else if (a == "--bipartite_index_save_path") bipartite_index_save_file = next();
        else if (a == "--entry_seed") entry_seed = std::stoll(next());
        else if (a == "--evaluation_save_path") evaluation_save_path = next();
        else if (a == "--k") k = (uint32_t)std::stoul(next());
        else if (a == "-T" || a == "--num_threads") num_threads = (uint32_t)std::stoul(next());
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--gpus") gpus = std::stoi(next());
        else if (a == "--devices") {
            std::string v = next();
            for (size_t p0 = 0; p0 <= v.size();) {
                size_t p1 = v.find(',', p0);
                if (p1 == std::string::npos) p1 = v.size();
                device_list.push_back(std::stoi(v.substr(p0, p1 - p0)));
                p0 = p1 + 1;
            }
            if (!device_list.empty()) device = device_list[0];
        }
        else if (a == "--shards") shard_devices = split_ints(next());
        else if (a == "--visited") visited = next();
        else if (a == "--per_query") per_query = true;
        else if (a == "--rderr") want_rderr = true;
        else if (a == "--L_pq") {
            while (i + 1 < argc && argv[i + 1][0] != '-') L_vec.push_back((uint32_t)std::stoul(argv[++i]));
        } else {
            std::cerr << "unrecognised option '" << a << "'\n";
            return -1;
        }
    }
    // --bipartite_index_save_path: the run of tests/test_search_bipartite.cpp (Load + SearchBipartiteGraph,
    // CSV row L_pq,qps,avg_cmps,latency,recall,rderr,avg_hops :226-231) instead of the projection-graph search
    const bool bipartite = !bipartite_index_save_file.empty();
    if (bipartite) {
        projection_index_save_file = bipartite_index_save_file;
        want_rderr = true;
    }
    if (dist.empty() || base_data_file.empty() || query_file.empty() || gt_file.empty() ||
        projection_index_save_file.empty() || L_vec.empty()) {
        std::cerr << "missing required option\n";
        usage();
        return -1;
    }
    try {
        uint32_t base_num = 0, base_dim = 0;
        if (shard_devices.empty()) efanna2e::load_meta<float>(base_data_file.c_str(), base_num, base_dim);
        efanna2e::Parameters parameters;
        parameters.Set<uint32_t>("num_threads", num_threads);
        omp_set_num_threads(num_threads);
        uint32_t q_pts, q_dim;
        efanna2e::load_meta<float>(query_file.c_str(), q_pts, q_dim);
        float *query_data = nullptr;
        efanna2e::load_data<float>(query_file.c_str(), q_pts, q_dim, query_data);
        // load_data already strides rows at the padded dim; re-striding from that is data_align
        uint32_t padded = (q_dim + 7) / 8 * 8;
        float *aligned_query_data = efanna2e::data_align(query_data, q_pts, padded);
        q_dim = padded;
        // the query block stays where it is for the whole run: let the GPU read it in place (opt-in; the block itself is
        // ordinary memory, released below like the reference releases its own, tests/test_search_roargraph.cpp:246)
        const bool pinned = efanna2e::pin_block(aligned_query_data, (uint64_t)q_pts * q_dim * sizeof(float));

        uint32_t gt_pts, gt_dim;
        uint32_t *gt_ids = nullptr;
        float *gt_dists = nullptr;
        efanna2e::load_gt_meta<uint32_t>(gt_file.c_str(), gt_pts, gt_dim);
        efanna2e::load_gt_data_with_dist<uint32_t, float>(gt_file.c_str(), gt_pts, gt_dim, gt_ids, gt_dists);
        efanna2e::Metric dist_metric = efanna2e::INNER_PRODUCT;
        if (dist == "l2") {
            dist_metric = efanna2e::L2;
            std::cout << "Using l2 as distance metric" << std::endl;
        } else if (dist == "ip") {
            std::cout << "Using inner product as distance metric" << std::endl;
        } else if (dist == "cosine") {
            dist_metric = efanna2e::COSINE;
            std::cout << "Using cosine as distance metric" << std::endl;
        } else {
            std::cout << "Unknown distance type: " << dist << std::endl;
            return -1;
        }
        if (!shard_devices.empty()) {
            // ---- index-sharded run: same loop and CSV as below, one ShardedIndexBipartite instead of one IndexBipartite
            const std::vector<std::string> bases = split_names(base_data_file), graphs = split_names(projection_index_save_file);
            if (bases.size() != shard_devices.size() || graphs.size() != shard_devices.size()) {
                std::cerr << "--shards names " << shard_devices.size() << " devices: give as many base files and index files" << std::endl;
                return -1;
            }
            efanna2e::ShardedIndexBipartite sharded(q_dim, dist_metric, shard_devices);
            for (size_t s2 = 0; s2 < bases.size(); ++s2) {
                sharded.LoadSearchNeededData(s2, bases[s2].c_str());
                std::cout << "Load graph index: " << graphs[s2] << " (shard " << s2 << ", device " << shard_devices[s2] << ")" << std::endl;
                sharded.LoadProjectionGraph(s2, graphs[s2].c_str());
            }
            if (sharded.need_normalize)
                for (uint32_t i = 0; i < q_pts; i++) efanna2e::normalize<float>(aligned_query_data + (size_t)i * q_dim, q_dim);
            sharded.SetOption("visited_mode", visited == "exact" ? ROAR_VISITED_EXACT : visited == "lds" ? ROAR_VISITED_LDS : ROAR_VISITED_AUTO);
            std::cout << "Index sharded " << sharded.NumShards() << "-way over " << sharded.Info("ranks") << " device(s)" << std::endl;
            std::cout << "k: " << k << std::endl;
            std::vector<uint32_t> res((size_t)q_pts * k, 0), cmps_vec(q_pts, 0), hops_vec(q_pts, 0);
            std::vector<float> res_d((size_t)q_pts * k, 0.f);
            std::ofstream evaluation_out;
            if (!evaluation_save_path.empty()) evaluation_out.open(evaluation_save_path, std::ios::out);
            std::cout << "L_pq\t\tQPS\t\t\tavg_visited\tmean_latency\trecall@" << k << "\tavg_hops" << std::endl;
            for (uint32_t L_pq : L_vec) {
                if (k > L_pq) {
                    std::cout << "L_pq must greater or equal than k" << std::endl;
                    exit(1);
                }
                sharded.SearchRoarGraphBatch(aligned_query_data, std::min<uint32_t>(100, q_pts), k, L_pq, res.data(), res_d.data(),
                                             cmps_vec.data(), hops_vec.data());  // warm-up, :198-200
                auto start = std::chrono::high_resolution_clock::now();
                sharded.SearchRoarGraphBatch(aligned_query_data, q_pts, k, L_pq, res.data(), res_d.data(), cmps_vec.data(), hops_vec.data());
                auto end = std::chrono::high_resolution_clock::now();
                double diff_ms = std::chrono::duration_cast<std::chrono::microseconds>(end - start).count() / 1000.0;
                float qps = (float)((double)q_pts / (diff_ms / 1000.0));
                float recall = ComputeRecall(q_pts, k, gt_dim, res.data(), gt_ids);
                double avg_cmps = 0, avg_hops = 0;
                for (size_t i = 0; i < q_pts; ++i) avg_cmps += cmps_vec[i], avg_hops += hops_vec[i];
                avg_cmps /= q_pts, avg_hops /= q_pts;
                std::cout << L_pq << "\t\t" << qps << "\t\t" << (float)avg_cmps << "\t\t" << (float)(diff_ms / q_pts) << "\t\t" << recall
                          << "\t\t" << (float)avg_hops << std::endl;
                if (evaluation_out.is_open())
                    evaluation_out << L_pq << "," << qps << "," << (float)avg_cmps << "," << (float)(diff_ms / q_pts) << "," << recall << ","
                                   << (float)avg_hops << std::endl;
            }
            if (pinned) efanna2e::unpin_block(aligned_query_data);
            free(aligned_query_data);
            delete[] gt_ids;
            delete[] gt_dists;
            return 0;
        }
        {
            std::ifstream probe(projection_index_save_file, std::ios::binary);
            if (!probe.is_open()) {
                std::cout << (bipartite ? "bipartite_index_save_file index file does not exist."
                                        : "projection index file does not exist.")
                          << std::endl;
                return -1;
            }
        }
        efanna2e::IndexBipartite index(q_dim, base_num, dist_metric, nullptr, device);
        index.LoadSearchNeededData(base_data_file.c_str(), "");
        std::cout << "Load graph index: " << projection_index_save_file << std::endl;
        if (bipartite) {
            index.Load(projection_index_save_file.c_str());
            if (entry_seed >= 0) index.SetOption("bipartite_seed", entry_seed);
        } else {
            index.LoadProjectionGraph(projection_index_save_file.c_str());
        }
        if (index.need_normalize) {
            std::cout << "Normalizing query data" << std::endl;
            for (uint32_t i = 0; i < q_pts; i++) efanna2e::normalize<float>(aligned_query_data + (size_t)i * q_dim, q_dim);
        }
        if (!device_list.empty() && !bipartite) {
            index.Replicate(std::vector<int>(device_list.begin() + 1, device_list.end()));
            std::cout << "Index replicated on " << index.NumReplicas() << " devices, queries sharded" << std::endl;
        } else if (gpus != 1 && !bipartite) {  // replicate after load: one host thread + stream per device inside the library
            int have = 0;
            if (gpus <= 0) {
                while (true) {  // count devices by asking for an index on each
                    roar_index *probe = nullptr;
                    if (roar_create(&probe, 8, ROAR_METRIC_L2, have) != ROAR_OK) break;
                    roar_destroy(probe);
                    ++have;
                }
                gpus = std::max(1, have - device);
            }
            std::vector<int> extra;
            for (int g = 1; g < gpus; ++g) extra.push_back(device + g);
            index.Replicate(extra);
            std::cout << "Index replicated on " << index.NumReplicas() << " devices, queries sharded" << std::endl;
        }
        index.SetOption("visited_mode", visited == "exact" ? ROAR_VISITED_EXACT
                                        : visited == "lds" ? ROAR_VISITED_LDS : ROAR_VISITED_AUTO);
        if (visited == "exact") index.InitVisitedListPool(num_threads);

        std::cout << "k: " << k << std::endl;
        std::vector<uint32_t> res((size_t)q_pts * k, 0);
        std::vector<float> res_d((size_t)q_pts * k, 0.f);
        std::vector<std::vector<float>> res_dists(q_pts, std::vector<float>(k, 0.0));
        std::vector<uint32_t> cmps_vec(q_pts, 0), hops_vec(q_pts, 0);
        std::ofstream evaluation_out;
        if (!evaluation_save_path.empty()) evaluation_out.open(evaluation_save_path, std::ios::out);
        std::cout << "Using thread: " << num_threads << std::endl;
        std::cout << "L_pq\t\tQPS\t\t\tavg_visited\tmean_latency\trecall@" << k << "\tavg_hops" << std::endl;
        for (uint32_t L_pq : L_vec) {
            if (k > L_pq) {
                std::cout << "L_pq must greater or equal than k" << std::endl;
                exit(1);
            }
            parameters.Set<uint32_t>("L_pq", L_pq);
            // warm-up, :198-200 (100 queries)
            uint32_t warm = std::min<uint32_t>(100, q_pts);
            if (bipartite)
                index.SearchBipartiteGraphBatch(aligned_query_data, warm, k, L_pq, nullptr, 10, res.data(), res_d.data(),
                                                cmps_vec.data(), hops_vec.data());
            else
                index.SearchRoarGraphBatch(aligned_query_data, warm, k, L_pq, res.data(), res_d.data(), cmps_vec.data(),
                                           hops_vec.data());
            auto start = std::chrono::high_resolution_clock::now();
            if (per_query) {
#pragma omp parallel for schedule(dynamic, 1)
                for (size_t i = 0; i < q_pts; ++i) {
                    auto ret = bipartite ? index.SearchBipartiteGraph(aligned_query_data + i * q_dim, k, i, parameters,
                                                                      res.data() + i * k, res_dists[i])
                                         : index.SearchRoarGraph(aligned_query_data + i * q_dim, k, i, parameters,
                                                                 res.data() + i * k, res_dists[i]);
                    cmps_vec[i] = ret.first;
                    hops_vec[i] = ret.second;
                }
            } else if (bipartite) {
                index.SearchBipartiteGraphBatch(aligned_query_data, q_pts, k, L_pq, nullptr, 10, res.data(), res_d.data(),
                                                cmps_vec.data(), hops_vec.data());
            } else {
                index.SearchRoarGraphBatch(aligned_query_data, q_pts, k, L_pq, res.data(), res_d.data(),
                                           cmps_vec.data(), hops_vec.data());
            }
            auto end = std::chrono::high_resolution_clock::now();
            double diff_ms = std::chrono::duration_cast<std::chrono::microseconds>(end - start).count() / 1000.0;
            float qps = (float)((double)q_pts / (diff_ms / 1000.0));
            float recall = ComputeRecall(q_pts, k, gt_dim, res.data(), gt_ids);
            double avg_cmps = 0, avg_hops = 0;
            for (size_t i = 0; i < q_pts; ++i) avg_cmps += cmps_vec[i], avg_hops += hops_vec[i];
            avg_cmps /= q_pts;
            avg_hops /= q_pts;
            if (per_query)  // the per-query path fills res_dists[i]; flatten for the error metric
                for (size_t i = 0; i < q_pts; ++i) std::copy(res_dists[i].begin(), res_dists[i].begin() + k, res_d.begin() + i * k);
            double rderr = want_rderr ? ComputeRderr(gt_dists, gt_dim, res_d.data(), q_pts, k, dist_metric) : 0.0;
            if (bipartite) {  // column order of tests/test_search_bipartite.cpp:224-231
                std::cout << L_pq << "\t\t" << qps << "\t\t" << (float)avg_cmps << "\t\t" << (float)(diff_ms / q_pts)
                          << "\t\t" << recall << "\t\t" << rderr << "\t\t" << (float)avg_hops << std::endl;
                if (evaluation_out.is_open())
                    evaluation_out << L_pq << "," << qps << "," << (float)avg_cmps << "," << (float)(diff_ms / q_pts) << ","
                                   << recall << "," << rderr << "," << (float)avg_hops << std::endl;
                continue;
            }
            std::cout << L_pq << "\t\t" << qps << "\t\t" << (float)avg_cmps << "\t\t" << (float)(diff_ms / q_pts) << "\t\t"
                      << recall << "\t\t" << (float)avg_hops;
            if (want_rderr) std::cout << "\t\t" << rderr;
            std::cout << std::endl;
            if (evaluation_out.is_open()) {
                evaluation_out << L_pq << "," << qps << "," << (float)avg_cmps << "," << (float)(diff_ms / q_pts) << ","
                               << recall << "," << (float)avg_hops;
                if (want_rderr) evaluation_out << "," << rderr;
                evaluation_out << std::endl;
            }
        }
        if (pinned) efanna2e::unpin_block(aligned_query_data);
        free(aligned_query_data);
        delete[] gt_ids;
        delete[] gt_dists;
    } catch (const std::exception &ex) {
        std::cerr << ex.what() << '\n';
        return -1;
    }
    return 0;
}
