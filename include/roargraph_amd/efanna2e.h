// efanna2e.h -- header-only C++ facade over the C ABI (roargraph_amd.h), source compatible with the
// members of the reference's headers that its search harness touches
// (tests/test_search_roargraph.cpp:118-209), so a harness written against
//   include/index_bipartite.h, include/efanna2e/{distance,parameters,util}.h
// compiles against this file unchanged and runs the search on an MI355X.
//
// Own implementation; each item cites the reference declaration it mirrors.
#ifndef ROARGRAPH_AMD_EFANNA2E_H
#define ROARGRAPH_AMD_EFANNA2E_H

#include <algorithm>
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
// Same names, arguments, messages and padding rule as the reference's free functions; the implementation is this
// library's: a file is sized once, read with one sized read and re-strided in memory.  The blocks are ordinary
// malloc-compatible memory (delete[] / free / host_free all release them, as with the reference's memalign blocks);
// pin_block / unpin_block are the opt-in for a query block the GPU should read in place.
static const unsigned kDataAlignFactor = 8;  // DATA_ALIGN_FACTOR with AVX, util.h:38-48

namespace detail {
inline uint64_t round_up_dim(uint64_t d) { return (d + kDataAlignFactor - 1) / kDataAlignFactor * kDataAlignFactor; }

struct FbinHeader {
    uint32_t n = 0, d = 0;
    uint64_t payload = 0;  // bytes after the two header words
};

// opens, reads `u32 n; u32 d`, sizes the payload; the reference prints and exits on an unopenable file (util.h:109-112)
inline FbinHeader open_fbin(const char *filename, std::ifstream &in) {
    in.open(filename, std::ios::binary | std::ios::ate);
    if (!in.is_open()) {
        std::cout << "open file error" << std::endl;
        exit(-1);
    }
    FbinHeader h;
    const uint64_t fsize = (uint64_t)in.tellg();
    in.seekg(0, std::ios::beg);
    in.read(reinterpret_cast<char *>(&h.n), 4);
    in.read(reinterpret_cast<char *>(&h.d), 4);
    h.payload = fsize >= 8 ? fsize - 8 : 0;
    return h;
}

// Blocks the reference-named loaders hand out are malloc-compatible, as the reference's are (memalign, util.h:64,196):
// its harness releases them with delete[] / free (tests/test_search_roargraph.cpp:242-247) and so may a ported caller.
inline void *host_alloc(uint64_t bytes) {
    void *p = aligned_alloc(64, (size_t)((bytes + 63) / 64 * 64));
    if (!p) throw std::bad_alloc();
    return p;
}
inline void host_free(void *p) { free(p); }

// rows of `d` values -> rows of `stride` values, zero filled
template <typename T>
inline void restride(const T *src, uint64_t n, uint64_t d, T *dst, uint64_t stride) {
    for (uint64_t i = 0; i < n; ++i) {
        T *row = dst + i * stride;
        std::copy(src + i * d, src + (i + 1) * d, row);
        std::fill(row + d, row + stride, T(0));
    }
}
}  // namespace detail

inline void host_free(void *block) { detail::host_free(block); }

// Opt-in fast path for a query block that the caller keeps alive and unchanged across searches: page-lock it and map it
// into the GPU's address space (hipHostRegister), so that IndexBipartite::SearchRoarGraphBatch reads the rows in place --
// no staging copy, no host-to-device copy in front of the kernel.  unpin_block BEFORE the block is released.  Returns
// false (and changes nothing) when there is no GPU runtime or the block cannot be locked; searching it still works.
inline bool pin_block(void *block, uint64_t bytes) { return roar_host_register(block, bytes) == ROAR_OK; }
inline void unpin_block(void *block) { (void)roar_host_unregister(block); }

// util.h:106-127: header of an fbin / ibin file; (payload / dim / sizeof(T)) must equal the point count
template <typename T>
void load_meta(const char *filename, unsigned &points_num, unsigned &dim) {
    std::ifstream in;
    const detail::FbinHeader h = detail::open_fbin(filename, in);
    points_num = h.n;
    dim = h.d;
    const uint32_t contained = h.d ? (uint32_t)(h.payload / h.d / sizeof(T)) : 0u;
    if (contained != h.n) {
        std::cerr << "Data file size wrong! Get points " << contained << " but should have " << h.n << std::endl;
        throw std::runtime_error("Data file size wrong!");
    }
}

// util.h:84-105: ground truth = an id block followed by a distance block, i.e. twice the payload of load_meta
template <typename T>
void load_gt_meta(const char *filename, unsigned &points_num, unsigned &dim) {
    std::ifstream in;
    const detail::FbinHeader h = detail::open_fbin(filename, in);
    points_num = h.n;
    dim = h.d;
    const uint32_t contained = h.d ? (uint32_t)(h.payload / h.d / sizeof(T)) : 0u;
    if (contained != 2u * h.n) throw std::runtime_error("Data file size wrong!");
}

// util.h:179-211: the payload as rows strided at the dimension rounded up to 8, zero filled
template <typename T>
void load_data(const char *filename, uint32_t &points_num, uint32_t &dim, T *&data) {
    std::ifstream in;
    const detail::FbinHeader h = detail::open_fbin(filename, in);
    const uint64_t n = points_num, d = dim, stride = detail::round_up_dim(d);
    if (h.payload != n * d * sizeof(T)) throw std::runtime_error("Data file size wrong!");
    std::vector<T> raw((size_t)(n * d));
    in.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)(n * d * sizeof(T)));
    if ((uint64_t)in.gcount() != n * d * sizeof(T)) throw std::runtime_error("Data file size wrong!");
    data = static_cast<T *>(detail::host_alloc(n * stride * sizeof(T)));
    detail::restride(raw.data(), n, d, data, stride);
}

// util.h:37-75: rows of `dim` floats -> rows of the padded dimension; dim is updated, the input block released
inline float *data_align(float *data_ori, unsigned point_num, unsigned &dim) {
    const uint64_t n = point_num, d = dim, stride = detail::round_up_dim(d);
    float *out = static_cast<float *>(detail::host_alloc(n * stride * sizeof(float)));
    detail::restride(data_ori, n, d, out, stride);
    detail::host_free(data_ori);
    dim = (unsigned)stride;
    return out;
}

// util.h:129-155: ids block then distances block, both points_num x dim (new[] blocks, as in the reference)
template <typename T, typename T2>
void load_gt_data_with_dist(const char *filename, uint32_t &points_num, uint32_t &dim, T *&data, T2 *&res_dists) {
    std::ifstream in;
    const detail::FbinHeader h = detail::open_fbin(filename, in);
    const uint64_t cells = (uint64_t)points_num * dim;
    if (h.payload != cells * (sizeof(T) + sizeof(T2))) throw std::runtime_error("Data file size wrong!");
    data = new T[(size_t)cells];
    res_dists = new T2[(size_t)cells];
    in.read(reinterpret_cast<char *>(data), (std::streamsize)(cells * sizeof(T)));
    in.read(reinterpret_cast<char *>(res_dists), (std::streamsize)(cells * sizeof(T2)));
    if (!in) throw std::runtime_error("Data file size wrong!");
}

// util.h:214-225: x / |x|, the norm summed sequentially in float
template <typename T>
inline void normalize(T *arr, const size_t dim) {
    float norm2 = 0.0f;
    for (const T *p = arr; p != arr + dim; ++p) norm2 += (float)(*p) * (float)(*p);
    const float norm = std::sqrt(norm2);
    for (T *p = arr; p != arr + dim; ++p) *p = (T)(*p / norm);
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
        for (roar_index *r : replicas_) roar_destroy(r);
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
    // With replicas (Replicate) the block is split by query over the devices: replica r takes queries
    // [r * nq / g, (r + 1) * nq / g) and writes those rows of ids / dists / cmps / hops (no collective).
    void SearchRoarGraphBatch(const float *queries, size_t nq, size_t k, uint32_t L_pq, unsigned *ids, float *dists,
                              uint32_t *cmps, uint32_t *hops) {
        if (replicas_.empty()) {
            check(roar_search_batch(handle(), queries, nq, (uint32_t)dimension_, (uint32_t)k, L_pq, ids, dists, cmps, hops,
                                    nullptr));
            return;
        }
        std::vector<roar_index *> all(1, handle());
        all.insert(all.end(), replicas_.begin(), replicas_.end());
        check(roar_search_batch_multi(all.data(), (uint32_t)all.size(), queries, nq, (uint32_t)dimension_, (uint32_t)k, L_pq,
                                      ids, dists, cmps, hops, nullptr));
    }
    // New: the index replicated on further devices (BASELINE configs[3]: index replicated, queries sharded).  Call
    // after the base rows and the projection graph are loaded; options set before are copied, later SetOption calls
    // reach every replica.
    void Replicate(const std::vector<int> &devices) {
        for (int d : devices) {
            roar_index *r = nullptr;
            check(roar_clone_to_device(handle(), d, &r));
            replicas_.push_back(r);
        }
    }
    size_t NumReplicas() const { return replicas_.size() + 1; }

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
    void SetOption(const char *key, int64_t v) {
        check(roar_set_option(handle(), key, v));
        for (roar_index *r : replicas_) check(roar_set_option(r, key, v));
    }
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
    std::vector<roar_index *> replicas_;
    std::vector<uint32_t> learn_base_knn_;  // index_bipartite.h:153
    uint32_t knn_n_ = 0, knn_k_ = 0;
};

// New (BASELINE configs[4]): the base set cut into contiguous id ranges, one RoarGraph per range
// (local ids), shard s on HIP device devices[s].  Same member names as IndexBipartite, with the shard number in front
// where a file belongs to one shard; SearchRoarGraphBatch runs every query on every shard and returns the merge of the
// per-shard lists (global ids) after ONE RCCL all-gather (roar_search_batch_sharded).
class ShardedIndexBipartite {
   public:
    ShardedIndexBipartite(const size_t dimension, Metric m, const std::vector<int> &devices) : dimension_(dimension) {
        if (m == COSINE) need_normalize = true;
        check(roar_sharded_create(&h_, (uint32_t)dimension, (int)m, devices.data(), (uint32_t)devices.size()));
        n_shards_ = devices.size();
    }
    ~ShardedIndexBipartite() {
        if (h_) roar_sharded_destroy(h_);
    }
    ShardedIndexBipartite(const ShardedIndexBipartite &) = delete;
    ShardedIndexBipartite &operator=(const ShardedIndexBipartite &) = delete;
    size_t NumShards() const { return n_shards_; }
    void LoadSearchNeededData(size_t s, const char *base_file) { check(roar_load_base_fbin(shard(s), base_file)); }
    void LoadProjectionGraph(size_t s, const char *filename) { check(roar_load_graph_file(shard(s), filename)); }
    void SetIdBase(size_t s, uint64_t first_global_id) { check(roar_sharded_set_id_base(h_, (uint32_t)s, first_global_id)); }
    void SetOption(const char *key, int64_t v) {
        for (size_t s = 0; s < n_shards_; ++s) check(roar_set_option(shard(s), key, v));
    }
    void SearchRoarGraphBatch(const float *queries, size_t nq, size_t k, uint32_t L_pq, unsigned *ids, float *dists,
                              uint32_t *cmps, uint32_t *hops) {
        check(roar_search_batch_sharded(h_, queries, nq, (uint32_t)dimension_, (uint32_t)k, L_pq, ids, dists, cmps, hops));
    }
    int64_t Info(const char *key) {
        int64_t v = 0;
        check(roar_sharded_info(h_, key, &v));
        return v;
    }
    roar_index *shard(size_t s) {
        roar_index *x = roar_sharded_shard(h_, (uint32_t)s);
        if (!x) throw std::invalid_argument(roar_last_error());
        return x;
    }
    bool need_normalize = false;

   private:
    static void check(int rc) {
        if (rc == ROAR_OK) return;
        const char *msg = roar_last_error();
        if (rc == ROAR_ERR_ARG) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }
    size_t dimension_, n_shards_ = 0;
    roar_sharded *h_ = nullptr;
};

}  // namespace efanna2e
#endif
