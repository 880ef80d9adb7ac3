#include <cstdint>
#include "gguf.h"

#include <cstring>
#include <fstream>

#include "visp_util.h"

namespace visp {
namespace {

enum { T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 };

struct reader {
    const uint8_t* p;
    const uint8_t* end;
    const char* path;

    void need(size_t n) const {
        if ((size_t)(end - p) < n) throw except("Failed to load GGUF model: %s (truncated file)", path);
    }
    template <typename T>
    T get() {
        need(sizeof(T));
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        uint64_t n = get<uint64_t>();
        need(n);
        std::string s((const char*)p, (size_t)n);
        p += n;
        return s;
    }
};

size_t type_size(int32_t t) {
    switch (t) {
        case GGML_F32: case GGML_I32: return 4;
        case GGML_F16: case GGML_I16: return 2;
        case GGML_I8: return 1;
        case GGML_I64: return 8;
        default: return 0;
    }
}

void read_scalar(reader& r, uint32_t type, gguf_value& v) {
    switch (type) {
        case T_U8: v.u = r.get<uint8_t>(); break;
        case T_I8: v.u = (uint64_t)(int64_t)r.get<int8_t>(); break;
        case T_U16: v.u = r.get<uint16_t>(); break;
        case T_I16: v.u = (uint64_t)(int64_t)r.get<int16_t>(); break;
        case T_U32: v.u = r.get<uint32_t>(); break;
        case T_I32: v.u = (uint64_t)(int64_t)r.get<int32_t>(); break;
        case T_F32: v.f = r.get<float>(); break;
        case T_BOOL: v.u = r.get<uint8_t>(); break;
        case T_U64: v.u = r.get<uint64_t>(); break;
        case T_I64: v.u = (uint64_t)r.get<int64_t>(); break;
        case T_F64: v.f = r.get<double>(); break;
        case T_STR: v.s = r.str(); break;
        default: throw except("Failed to load GGUF model: %s (bad value type %u)", r.path, type);
    }
}

} // namespace

model_file model_load(const char* filepath, bool header_only) {
    model_file f;
    f.path = filepath;
    std::ifstream in(filepath, std::ios::binary | std::ios::ate);
    if (!in) throw except("Failed to load GGUF model: %s", filepath);
    std::streamsize size = in.tellg();
    in.seekg(0);
    // header_only still needs the whole header; headers are small, read up to 16 MiB then
    std::streamsize want = header_only ? std::min<std::streamsize>(size, 16 << 20) : size;
    f.buffer.resize((size_t)want);
    if (!in.read((char*)f.buffer.data(), want)) throw except("Failed to load GGUF model: %s", filepath);

    reader r{f.buffer.data(), f.buffer.data() + f.buffer.size(), filepath};
    r.need(4);
    if (memcmp(r.p, "GGUF", 4) != 0) throw except("Failed to load GGUF model: %s (bad magic)", filepath);
    r.p += 4;
    uint32_t version = r.get<uint32_t>();
    if (version != 2 && version != 3) throw except("Failed to load GGUF model: %s (version %u)", filepath, version);
    uint64_t n_tensors = r.get<uint64_t>();
    uint64_t n_kv = r.get<uint64_t>();
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string key = r.str();
        gguf_value v;
        v.type = r.get<uint32_t>();
        if (v.type == T_ARR) {
            v.arr_type = r.get<uint32_t>();
            uint64_t n = r.get<uint64_t>();
            for (uint64_t j = 0; j < n; ++j) {
                gguf_value e;
                read_scalar(r, v.arr_type, e);
                if (v.arr_type == T_STR) v.arr_s.push_back(std::move(e.s));
                else if (v.arr_type == T_F32 || v.arr_type == T_F64) v.arr_f.push_back(e.f);
                else v.arr_i.push_back((int64_t)e.u);
            }
        } else {
            read_scalar(r, v.type, v);
        }
        f.kv.emplace(std::move(key), std::move(v));
    }
    std::vector<uint64_t> offsets(n_tensors);
    f.tensors.resize(n_tensors);
    for (uint64_t i = 0; i < n_tensors; ++i) {
        gguf_tensor& t = f.tensors[i];
        t.name = r.str();
        uint32_t nd = r.get<uint32_t>();
        if (nd > 4) throw except("Failed to load GGUF model: %s (tensor %s has %u dims)", filepath, t.name.c_str(), nd);
        // untrusted 64-bit header fields: every dimension >= 1 and the element / byte counts overflow-checked
        uint64_t n_elem = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            const uint64_t ne = r.get<uint64_t>();
            if (ne == 0 || ne > (uint64_t)INT64_MAX || n_elem > (uint64_t)INT64_MAX / ne)
                throw except("Failed to load GGUF model: %s (tensor %s: bad dimension %u)", filepath, t.name.c_str(), d);
            n_elem *= ne;
            t.ne[d] = (int64_t)ne;
        }
        t.type = (int32_t)r.get<uint32_t>();
        offsets[i] = r.get<uint64_t>();
        size_t ts = type_size(t.type);
        if (ts == 0) throw except("Failed to load GGUF model: %s (tensor %s: unsupported type %d)", filepath, t.name.c_str(), t.type);
        if (n_elem > (uint64_t)SIZE_MAX / ts) throw except("Failed to load GGUF model: %s (tensor %s: size overflows)", filepath, t.name.c_str());
        t.n_bytes = (size_t)n_elem * ts;
        f.index.emplace(t.name, (int)i);
    }
    uint64_t align = 32;
    if (const gguf_value* a = f.find_key("general.alignment")) align = a->u ? a->u : 32;
    if (align > (1u << 20) || (align & (align - 1)) != 0) throw except("Failed to load GGUF model: %s (bad general.alignment)", filepath);
    size_t base = (size_t)(r.p - f.buffer.data());
    base = (base + align - 1) / align * align;
    if (!header_only) {
        if (base > f.buffer.size()) throw except("Failed to load GGUF model: %s (truncated before the tensor data)", filepath);
        const size_t avail = f.buffer.size() - base;
        for (uint64_t i = 0; i < n_tensors; ++i) {
            gguf_tensor& t = f.tensors[i];
            // no sum of untrusted values: offset <= avail, size <= avail - offset, offset aligned
            if (offsets[i] > avail || t.n_bytes > avail - (size_t)offsets[i] || offsets[i] % align != 0)
                throw except("Failed to load GGUF model: %s (tensor %s out of bounds)", filepath, t.name.c_str());
            t.data = f.buffer.data() + base + (size_t)offsets[i];
        }
    }
    return f;
}

const gguf_value* model_file::find_key(std::string_view name) const {
    auto it = kv.find(name);
    return it == kv.end() ? nullptr : &it->second;
}
const gguf_value& model_file::key(std::string_view name) const {
    if (const gguf_value* v = find_key(name)) return *v;
    throw except("Can't find key '%.*s' in model file %s", (int)name.size(), name.data(), path.c_str());
}
std::string_view model_file::get_string(std::string_view name) const {
    const gguf_value& v = key(name);
    if (v.type != T_STR) throw except("Key '%.*s' is not a string in %s", (int)name.size(), name.data(), path.c_str());
    return v.s;
}
int model_file::get_int(std::string_view name) const {
    const gguf_value& v = key(name);
    if (v.type != T_I32) throw except("Key '%.*s' is not int32 in %s", (int)name.size(), name.data(), path.c_str());
    return (int)(int64_t)v.u;
}
uint32_t model_file::get_uint32(std::string_view name) const {
    const gguf_value& v = key(name);
    if (v.type != T_U32) throw except("Key '%.*s' is not uint32 in %s", (int)name.size(), name.data(), path.c_str());
    return (uint32_t)v.u;
}
void model_file::get_array(std::string_view name, int* out, size_t n) const {
    const gguf_value& v = key(name);
    if (v.type != T_ARR || v.arr_i.size() != n)
        throw except("Array size mismatch for key '%.*s' in model file %s", (int)name.size(), name.data(), path.c_str());
    if (v.arr_type != T_I32)
        throw except("Array type mismatch for key '%.*s' in model file %s, expected int32", (int)name.size(), name.data(), path.c_str());
    for (size_t i = 0; i < n; ++i) out[i] = (int)v.arr_i[i];
}
std::string_view model_file::arch() const { return get_string("general.architecture"); }
int32_t model_file::float_type() const {
    if (const gguf_value* v = find_key("general.file_type"))
        if (v->type == T_U32) return (int32_t)v->u;
    return GGML_TYPE_NONE;
}
tensor_data_layout model_file::tensor_layout() const {
    std::string k = std::string(arch()) + ".tensor_data_layout";
    if (const gguf_value* v = find_key(k)) {
        if (v->s == "cwhn") return layout_cwhn;
        if (v->s == "whcn") return layout_whcn;
    }
    return layout_unknown;
}
std::vector<int32_t> model_file::conv2d_weights() const {
    std::string k = std::string(arch()) + ".conv2d_weights";
    std::vector<int32_t> out;
    if (const gguf_value* v = find_key(k))
        if (v->type == T_ARR && v->arr_type == T_I32)
            for (int64_t i : v->arr_i) out.push_back((int32_t)i);
    return out;
}
const gguf_tensor* model_file::find(std::string_view name) const {
    auto it = index.find(name);
    return it == index.end() ? nullptr : &tensors[it->second];
}
const gguf_tensor& model_file::tensor(std::string_view name) const {
    if (const gguf_tensor* t = find(name)) return *t;
    throw except("tensor not found: %.*s", (int)name.size(), name.data());
}

} // namespace visp
