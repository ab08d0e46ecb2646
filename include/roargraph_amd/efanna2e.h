// efanna2e.h -- header-only C++ facade over the C ABI (roargraph_amd.h), source compatible with the
// members of the reference's headers that its search harness touches
// (tests/test_search_roargraph.cpp:118-209), so a harness written against
//   include/index_bipartite.h, include/efanna2e/{distance,parameters,util}.h
// compiles against this file unchanged and runs the search on an MI355X.
//
// Own implementation; each item cites the reference declaration it mirrors.
#ifndef ROARGRAPH_AMD_EFANNA2E_H
#define ROARGRAPH_AMD_EFANNA2E_H

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../roargraph_amd.h"

namespace efanna2e {

// include/efanna2e/distance.h:15
enum Metric { L2 = 0, INNER_PRODUCT = 1, FAST_L2 = 2, PQ = 3, COSINE = 4 };

// include/efanna2e/parameters.h:15-57
class Parameters {
   public:
    template <typename T>
    inline void Set(const std::string &name, const T &value) {
        std::stringstream ss;
        ss << value;
        params_[name] = ss.str();
    }
    template <typename T>
    inline T Get(const std::string &name) const {
        auto it = params_.find(name);
        if (it == params_.end()) throw std::invalid_argument("Invalid parameter name.");
        std::stringstream ss(it->second);
        T value;
        if (!(ss >> value) || !ss.eof()) {
            std::stringstream err;
            err << "Failed to convert value '" << it->second << "' to type: " << typeid(value).name();
            throw std::runtime_error(err.str());
        }
        return value;
    }
    template <typename T>
    inline T Get(const std::string &name, const T &default_value) const {
        try {
            return Get<T>(name);
        } catch (const std::invalid_argument &) {
            return default_value;
        }
    }

   private:
    std::unordered_map<std::string, std::string> params_;
};

// ---- include/efanna2e/util.h ---------------------------------------------------------------------
static const unsigned kDataAlignFactor = 8;  // DATA_ALIGN_FACTOR with AVX, util.h:38-48

// util.h:106-127
template <typename T>
void load_meta(const char *filename, unsigned &points_num, unsigned &dim) {
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) {
        std::cout << "open file error" << std::endl;
        exit(-1);
    }
    in.read((char *)&points_num, 4);
    in.read((char *)&dim, 4);
    in.seekg(0, std::ios::end);
    size_t fsize = (size_t)in.tellg();
    uint32_t contained = dim ? (unsigned)((fsize - sizeof(uint32_t) * 2) / dim / sizeof(T)) : 0;
    if (points_num != contained) {
        std::cerr << "Data file size wrong! Get points " << contained << " but should have " << points_num << std::endl;
        throw std::runtime_error("Data file size wrong!");
    }
}

// util.h:84-105 (ids block + distance block: twice the payload)
template <typename T>
void load_gt_meta(const char *filename, unsigned &points_num, unsigned &dim) {
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) {
        std::cout << "open file error" << std::endl;
        exit(-1);
    }
    in.read((char *)&points_num, 4);
    in.read((char *)&dim, 4);
    in.seekg(0, std::ios::end);
    size_t fsize = (size_t)in.tellg();
    uint32_t contained = dim ? (unsigned)((fsize - sizeof(uint32_t) * 2) / dim / sizeof(T)) : 0;
    if (points_num * 2 != contained) throw std::runtime_error("Data file size wrong!");
}

// util.h:179-211: rows strided at the padded dimension, zero filled, 64-byte aligned block
template <typename T>
void load_data(const char *filename, uint32_t &points_num, uint32_t &dim, T *&data) {
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) {
        std::cout << "open file error" << std::endl;
        exit(-1);
    }
    in.seekg(sizeof(uint32_t) * 2, std::ios::beg);
    uint64_t pts = points_num, d = dim;
    uint64_t new_dim = (d + kDataAlignFactor - 1) / kDataAlignFactor * kDataAlignFactor;
    data = (T *)aligned_alloc(64, ((pts * new_dim * sizeof(T) + 63) / 64) * 64);
    for (size_t i = 0; i < points_num; i++) {
        in.read((char *)(data + i * new_dim), d * sizeof(T));
        memset(data + i * new_dim + d, 0, (new_dim - d) * sizeof(T));
    }
    if ((size_t)in.tellg() != pts * d * sizeof(T) + sizeof(uint32_t) * 2) {
        free(data);
        data = nullptr;
        throw std::runtime_error("Data file size wrong!");
    }
}

// util.h:37-75: re-stride to the padded dimension, update dim, release the input block
inline float *data_align(float *data_ori, unsigned point_num, unsigned &dim) {
    uint64_t pts = point_num, d = dim;
    uint64_t new_dim = (d + kDataAlignFactor - 1) / kDataAlignFactor * kDataAlignFactor;
    float *data_new = (float *)aligned_alloc(64, ((pts * new_dim * sizeof(float) + 63) / 64) * 64);
    for (uint64_t i = 0; i < pts; i++) {
        memcpy(data_new + i * new_dim, data_ori + i * d, d * sizeof(float));
        memset(data_new + i * new_dim + d, 0, (new_dim - d) * sizeof(float));
    }
    dim = (unsigned)new_dim;
    free(data_ori);
    return data_new;
}

// util.h:129-155
template <typename T, typename T2>
void load_gt_data_with_dist(const char *filename, uint32_t &points_num, uint32_t &dim, T *&data, T2 *&res_dists) {
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) {
        std::cout << "open file error" << std::endl;
        exit(-1);
    }
    in.seekg(sizeof(uint32_t) * 2, std::ios::beg);
    data = new T[(size_t)points_num * dim];
    res_dists = new T2[(size_t)points_num * dim];
    in.read((char *)data, (size_t)points_num * dim * sizeof(T));
    in.read((char *)res_dists, (size_t)points_num * dim * sizeof(T2));
    if ((size_t)in.tellg() != (size_t)points_num * dim * sizeof(T) * 2 + sizeof(uint32_t) * 2)
        throw std::runtime_error("Data file size wrong!");
}

// util.h:214-225
template <typename T>
inline void normalize(T *arr, const size_t dim) {
    float sum = 0.0f;
    for (size_t i = 0; i < dim; i++) sum += arr[i] * arr[i];
    sum = std::sqrt(sum);
    for (size_t i = 0; i < dim; i++) arr[i] = (T)(arr[i] / sum);
}

// ---- include/index_bipartite.h (search path only) -------------------------------------------------
class Index;  // the `initializer` argument is unused on the search path

class IndexBipartite {
   public:
    // index_bipartite.h:27, src/index_bipartite.cpp:30-38.  `dimension` may be the padded query
    // dimension (the harness passes q_dim after data_align): base files are matched on the
    // unpadded dimension rounded up to 8.
    explicit IndexBipartite(const size_t dimension, const size_t n, Metric m, Index *initializer, int device = 0)
        : dimension_(dimension), nd_(n), metric_(m), device_(device) {
        (void)initializer;
        if (m == COSINE) need_normalize = true;
    }
    ~IndexBipartite() {
        if (h_) roar_destroy(h_);
    }
    IndexBipartite(const IndexBipartite &) = delete;
    IndexBipartite &operator=(const IndexBipartite &) = delete;

    // index_bipartite.h:62-64 -> LoadVectorData, src/index_bipartite.cpp:2661-2692
    void LoadSearchNeededData(const char *base_file, const char *sampled_query_file) {
        (void)sampled_query_file;
        unsigned n = 0, d = 0;
        load_meta<float>(base_file, n, d);
        if ((d + 7) / 8 * 8 != (dimension_ + 7) / 8 * 8) throw std::runtime_error("base and query dimension mismatch");
        ensure(d);
        check(roar_load_base_fbin(h_, base_file));
        nd_ = n;
    }
    // index_bipartite.h:105, src/index_bipartite.cpp:2097-2117
    void LoadProjectionGraph(const char *filename) {
        ensure((unsigned)dimension_);
        check(roar_load_graph_file(h_, filename));
        int64_t ep = 0, ne = 0, n = 0;
        roar_get_info(h_, "ep", &ep);
        roar_get_info(h_, "nedges", &ne);
        roar_get_info(h_, "n", &n);
        std::cout << "Projection graph, ep: " << ep << std::endl;
        if (n) std::cout << "Projection graph, avg_degree: " << (float)ne / (float)n << std::endl;
    }
    // index_bipartite.h:109, src/index_bipartite.cpp:2622-2639: u32 n; u32 K; ids[n*K]
    void LoadLearnBaseKNN(const char *filename) {
        std::ifstream in(filename, std::ios::binary);
        if (!in.is_open()) throw std::runtime_error("learn base knn file error");
        uint32_t npts = 0, k_dim = 0;
        in.read((char *)&npts, sizeof(npts));
        in.read((char *)&k_dim, sizeof(k_dim));
        std::cout << "learn base knn npts: " << npts << ", k_dim: " << k_dim << std::endl;
        learn_base_knn_.resize((size_t)npts * k_dim);
        in.read((char *)learn_base_knn_.data(), (std::streamsize)learn_base_knn_.size() * 4);
        if (!in || npts == 0 || k_dim == 0) throw std::runtime_error("learn base knn file error");
        knn_n_ = npts;
        knn_k_ = k_dim;
    }
    // index_bipartite.h:38-39, src/index_bipartite.cpp:143-233 (reference shape on the device; the training
    // queries enter only through the kNN lists, as in the reference)
    void BuildRoarGraph(size_t n_sq, const float *sq_data, size_t n_bp, const float *bp_data, const Parameters &parameters) {
        (void)n_sq;
        (void)sq_data;
        ensure((unsigned)dimension_);
        check(roar_set_base(h_, bp_data, n_bp, (uint32_t)((dimension_ + 7) / 8 * 8)));
        nd_ = n_bp;
        check(roar_build_roargraph(h_, learn_base_knn_.data(), knn_n_, knn_k_, parameters.Get<uint32_t>("M_sq"),
                                   parameters.Get<uint32_t>("M_pjbp"), parameters.Get<uint32_t>("L_pjpq")));
    }
    // index_bipartite.h:102, src/index_bipartite.cpp:2606-2619
    void SaveProjectionGraph(const char *filename) { check(roar_save_graph_file(handle(), filename)); }
    // index_bipartite.h:133
    void InitVisitedListPool(uint32_t num_threads) { check(roar_init_visited_pool(handle(), num_threads)); }

    // index_bipartite.h:100-101, src/index_bipartite.cpp:2311-2420.  Thread-safe: each call is a
    // one-query launch.  Use SearchRoarGraphBatch to feed the GPU.
    std::pair<uint32_t, uint32_t> SearchRoarGraph(const float *query, size_t k, size_t &qid, const Parameters &parameters,
                                                  unsigned *indices, std::vector<float> &res_dists) {
        (void)qid;
        uint32_t L_pq = parameters.Get<uint32_t>("L_pq");
        uint32_t cmps = 0, hops = 0;
        if (res_dists.size() < k) res_dists.resize(k);
        check(roar_search_batch(handle(), query, 1, (uint32_t)dimension_, (uint32_t)k, L_pq, indices, res_dists.data(),
                                &cmps, &hops, nullptr));
        return std::make_pair(cmps, hops);
    }

    // New: the whole query block in one launch (what `#pragma omp parallel for` over SearchRoarGraph
    // does in tests/test_search_roargraph.cpp:203-209).  queries: nq rows, `dimension` floats apart.
    void SearchRoarGraphBatch(const float *queries, size_t nq, size_t k, uint32_t L_pq, unsigned *ids, float *dists,
                              uint32_t *cmps, uint32_t *hops) {
        check(roar_search_batch(handle(), queries, nq, (uint32_t)dimension_, (uint32_t)k, L_pq, ids, dists, cmps, hops,
                                nullptr));
    }

    // ---- bipartite two-hop variant (the ablation baseline) ---------------------------------------
    // index_bipartite.h:35-36, src/index_bipartite.cpp:42-141 -> qbaseNNbipartite :235-280
    void BuildBipartite(size_t n_sq, const float *sq_data, size_t n_bp, const float *bp_data, const Parameters &parameters) {
        (void)n_sq;
        (void)sq_data;
        ensure((unsigned)dimension_);
        check(roar_set_base(h_, bp_data, n_bp, (uint32_t)((dimension_ + 7) / 8 * 8)));
        nd_ = n_bp;
        check(roar_build_bipartite(h_, learn_base_knn_.data(), knn_n_, knn_k_, parameters.Get<uint32_t>("M_pjbp")));
    }
    // index_bipartite.h:53-54, src/index_bipartite.cpp:2045-2071
    void Save(const char *filename) { check(roar_save_bipartite_file(handle(), filename)); }
    void Load(const char *filename) {
        ensure((unsigned)dimension_);
        check(roar_load_bipartite_file(h_, filename));
    }
    // index_bipartite.h:137-138, src/index_bipartite.cpp:282-356.  The 10 entry points are drawn by the
    // library (the reference draws them from std::random_device, :287-294).
    std::pair<uint32_t, uint32_t> SearchBipartiteGraph(const float *query, size_t k, size_t &qid, const Parameters &parameters,
                                                       unsigned *indices, std::vector<float> &dists) {
        (void)qid;
        uint32_t L_pq = parameters.Get<uint32_t>("L_pq");
        uint32_t cmps = 0, hops = 0;
        if (dists.size() < k) dists.resize(k);
        check(roar_search_bipartite_batch(handle(), query, 1, (uint32_t)dimension_, (uint32_t)k, L_pq, nullptr, 10, indices,
                                          dists.data(), &cmps, &hops));
        return std::make_pair(cmps, hops);
    }
    // New: the whole query block in one launch (tests/test_search_bipartite.cpp:194-201).  init_ids: nq x n_init
    // base ids, or nullptr for library-drawn ones.
    void SearchBipartiteGraphBatch(const float *queries, size_t nq, size_t k, uint32_t L_pq, const uint32_t *init_ids,
                                   uint32_t n_init, unsigned *ids, float *dists, uint32_t *cmps, uint32_t *hops) {
        check(roar_search_bipartite_batch(handle(), queries, nq, (uint32_t)dimension_, (uint32_t)k, L_pq, init_ids, n_init,
                                          ids, dists, cmps, hops));
    }

    // exact kNN of `queries` among the loaded base rows (the producer of ground-truth / learn-base kNN files)
    void ExactKnn(const float *queries, size_t nq, size_t k, unsigned *ids, float *dists) {
        check(roar_exact_knn(handle(), queries, nq, (uint32_t)dimension_, (uint32_t)k, ids, dists));
    }
    void SetOption(const char *key, int64_t v) { check(roar_set_option(handle(), key, v)); }
    int64_t Info(const char *key) {
        int64_t v = 0;
        check(roar_get_info(handle(), key, &v));
        return v;
    }
    roar_index *handle() {
        if (!h_) throw std::runtime_error("index has no data loaded");
        return h_;
    }

    bool need_normalize = false;  // index_bipartite.h:145

   private:
    void ensure(unsigned file_dim) {
        if (h_) return;
        check(roar_create(&h_, file_dim, (int)metric_, device_));
    }
    static void check(int rc) {
        if (rc == ROAR_OK) return;
        const char *msg = roar_last_error();
        if (rc == ROAR_ERR_ARG) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);  // same texts as the reference's own throws
    }
    size_t dimension_, nd_;
    Metric metric_;
    int device_;
    roar_index *h_ = nullptr;
    std::vector<uint32_t> learn_base_knn_;  // index_bipartite.h:153
    uint32_t knn_n_ = 0, knn_k_ = 0;
};

}  // namespace efanna2e
#endif
