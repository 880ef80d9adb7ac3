// GGUF v2/v3 reader (own implementation; the reference uses ggml's gguf_* API:
// src/visp/ml.cpp:193-281, 435-445). Holds the whole file in memory like model_load
// (ml.cpp:206-217, gguf_init_from_file with no_alloc=false).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace visp {

enum ggml_type_id : int32_t { GGML_F32 = 0, GGML_F16 = 1, GGML_I8 = 24, GGML_I16 = 25, GGML_I32 = 26, GGML_I64 = 27, GGML_TYPE_NONE = -1 };
enum tensor_data_layout { layout_unknown, layout_whcn, layout_cwhn };

struct gguf_value {
    uint32_t type = 0;           // gguf value type id (0..12)
    uint64_t u = 0;              // integer / bool payload
    double f = 0;                // float payload
    std::string s;               // string payload
    uint32_t arr_type = 0;       // element type of arrays
    std::vector<int64_t> arr_i;  // integer arrays
    std::vector<double> arr_f;   // float arrays
    std::vector<std::string> arr_s;
};

struct gguf_tensor {
    std::string name;
    int32_t type = GGML_TYPE_NONE;
    int64_t ne[4] = {1, 1, 1, 1}; // ggml order: ne[0] contiguous
    const uint8_t* data = nullptr;
    size_t n_bytes = 0;
    int64_t n_elements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};

struct model_file {
    std::string path;
    std::vector<uint8_t> buffer;
    std::map<std::string, gguf_value, std::less<>> kv;
    std::vector<gguf_tensor> tensors;               // file order (indices used by conv2d_weights)
    std::map<std::string, int, std::less<>> index;

    int64_t n_tensors() const { return (int64_t)tensors.size(); }
    const gguf_value* find_key(std::string_view name) const;
    const gguf_value& key(std::string_view name) const;       // throws if missing (ml.cpp:223-229)
    std::string_view get_string(std::string_view name) const;
    int get_int(std::string_view name) const;                  // must be stored as i32 (ml.cpp:235-237)
    uint32_t get_uint32(std::string_view name) const;
    void get_array(std::string_view name, int* out, size_t n) const; // i32 array of exactly n (ml.cpp:243-254)
    std::string_view arch() const;                             // general.architecture
    int32_t float_type() const;                                // general.file_type u32, else NONE (ml.cpp:260-267)
    tensor_data_layout tensor_layout() const;                  // <arch>.tensor_data_layout (ml.cpp:269-281)
    std::vector<int32_t> conv2d_weights() const;               // <arch>.conv2d_weights (ml.cpp:435-445)
    const gguf_tensor* find(std::string_view name) const;
    const gguf_tensor& tensor(std::string_view name) const;    // throws "tensor not found"
};

model_file model_load(const char* filepath, bool header_only = false);

} // namespace visp
