// assets.cpp -- the data formats either side of the hot path:
//   * Wavefront OBJ -> de-indexed PNCV vertices with the asset baker's
//     conventions (asset-baker/asset_main.cpp:38-100),
//   * the reference's asset container (asset-lib/src/asset_loader.cpp:13-61)
//     for "MESH" (mesh_asset.cpp) and "TEXI" (texture_asset.cpp) files,
//   * an in-tree LZ4 block codec (the reference links lz4 >= 1.9.4; the block
//     format is restated from its published specification, lz4_Block_format.md).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_internal.h"

namespace crt {

// ------------------------------------------------------------------ LZ4 -----
// Block format: sequences of [token][literal-length ext][literals][offset LE16][match-length ext];
// token = (literal length << 4) | (match length - 4); 15 in a nibble means "add following bytes until != 255".
// End conditions: the last sequence is literals only; the last match starts >= 12 bytes before the end
// and the last 5 bytes are literals.
static int lz4_decompress(const uint8_t *src, int n, uint8_t *dst, int cap) {
    int ip = 0, op = 0;
    while (ip < n) {
        uint32_t token = src[ip++];
        uint32_t lit = token >> 4;
        if (lit == 15) { uint8_t s; do { if (ip >= n) return -1; s = src[ip++]; lit += s; } while (s == 255); }
        if ((int64_t)ip + lit > n || (int64_t)op + lit > cap) return -1;
        memcpy(dst + op, src + ip, lit); ip += (int)lit; op += (int)lit;
        if (ip >= n) break;                               // last sequence: literals only
        if (ip + 2 > n) return -1;
        uint32_t offset = src[ip] | ((uint32_t)src[ip + 1] << 8); ip += 2;
        if (offset == 0 || (int)offset > op) return -1;
        uint32_t ml = token & 15u;
        if (ml == 15) { uint8_t s; do { if (ip >= n) return -1; s = src[ip++]; ml += s; } while (s == 255); }
        ml += 4;
        if ((int64_t)op + ml > cap) return -1;
        for (uint32_t i = 0; i < ml; i++) { dst[op] = dst[op - (int)offset]; op++; }   // byte-wise: overlapping copies are the RLE case
    }
    return op;
}

static int lz4_bound(int n) { return n + n / 255 + 16; }

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

static int lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap) {
    if (cap < lz4_bound(n)) return -1;
    const int HASH_BITS = 16;
    std::vector<int> table(1u << HASH_BITS, -1);
    int ip = 0, anchor = 0, op = 0;
    const int mflimit = n - 12;           // a match may not start after this
    const int matchlimit = n - 5;         // and may not extend past this
    auto emit = [&](int lit_start, int lit_len, int offset, int mlen) {
        uint32_t lit_tok = lit_len >= 15 ? 15u : (uint32_t)lit_len;
        uint32_t ml_tok = mlen < 0 ? 0u : (mlen - 4 >= 15 ? 15u : (uint32_t)(mlen - 4));
        dst[op++] = (uint8_t)((lit_tok << 4) | ml_tok);
        if (lit_len >= 15) { int r = lit_len - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
        memcpy(dst + op, src + lit_start, (size_t)lit_len); op += lit_len;
        if (mlen < 0) return;
        dst[op++] = (uint8_t)(offset & 255); dst[op++] = (uint8_t)(offset >> 8);
        if (mlen - 4 >= 15) { int r = mlen - 4 - 15; while (r >= 255) { dst[op++] = 255; r -= 255; } dst[op++] = (uint8_t)r; }
    };
    while (ip <= mflimit) {
        uint32_t h = (rd32(src + ip) * 2654435761u) >> (32 - HASH_BITS);
        int cand = table[h];
        table[h] = ip;
        if (cand >= 0 && ip - cand <= 65535 && rd32(src + cand) == rd32(src + ip)) {
            int mlen = 4;
            while (ip + mlen < matchlimit && src[cand + mlen] == src[ip + mlen]) mlen++;
            emit(anchor, ip - anchor, ip - cand, mlen);
            ip += mlen; anchor = ip;
        } else ip++;
    }
    emit(anchor, n - anchor, 0, -1);
    return op;
}

// ------------------------------------------------------- tiny JSON reads ----
// The container's JSON is a flat object of numbers, strings and one number
// array (nlohmann::json::dump() of the maps in mesh_asset.cpp:68-92 /
// texture_asset.cpp:20-26), so key lookup by text search is sufficient.
static bool json_find(const std::string &js, const char *key, size_t &pos) {
    std::string pat = std::string("\"") + key + "\"";
    size_t k = js.find(pat);
    if (k == std::string::npos) return false;
    k = js.find(':', k + pat.size());
    if (k == std::string::npos) return false;
    pos = k + 1;
    while (pos < js.size() && (js[pos] == ' ' || js[pos] == '\t' || js[pos] == '\n')) pos++;
    return true;
}
static bool json_number(const std::string &js, const char *key, double &out) {
    size_t p; if (!json_find(js, key, p)) return false;
    out = strtod(js.c_str() + p, nullptr); return true;
}
static bool json_string(const std::string &js, const char *key, std::string &out) {
    size_t p; if (!json_find(js, key, p) || js[p] != '"') return false;
    size_t e = js.find('"', p + 1);
    if (e == std::string::npos) return false;
    out = js.substr(p + 1, e - p - 1); return true;
}
static std::string json_escape(const std::string &s) {
    std::string o;
    for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; }
    return o;
}

struct AssetFile { char type[4]; uint32_t version; std::string json; std::vector<uint8_t> blob; };

// asset_loader.cpp:37-61
static bool load_binaryfile(const std::string &path, AssetFile &f, std::string &err) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open " + path; return false; }
    uint32_t hdr[3];
    bool ok = fread(f.type, 1, 4, fp) == 4 && fread(hdr, 4, 3, fp) == 3;
    if (ok) {
        // the header's sizes are checked against what the file really holds before anything is allocated from them
        long here = ftell(fp), end = -1;
        if (here >= 0 && fseek(fp, 0, SEEK_END) == 0) end = ftell(fp);
        if (end < here || fseek(fp, here, SEEK_SET) != 0 || (uint64_t)hdr[1] + (uint64_t)hdr[2] > (uint64_t)(end - here)) {
            fclose(fp);
            err = "asset header of " + path + " claims more bytes than the file has";
            return false;
        }
        f.version = hdr[0];
        f.json.resize(hdr[1]); f.blob.resize(hdr[2]);
        ok = (hdr[1] == 0 || fread(&f.json[0], 1, hdr[1], fp) == hdr[1]) && (hdr[2] == 0 || fread(f.blob.data(), 1, hdr[2], fp) == hdr[2]);
    }
    fclose(fp);
    if (!ok) err = "truncated asset file " + path;
    return ok;
}
// asset_loader.cpp:13-34
static bool save_binaryfile(const std::string &path, const AssetFile &f, std::string &err) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { err = "cannot create " + path; return false; }
    uint32_t hdr[3] = {f.version, (uint32_t)f.json.size(), (uint32_t)f.blob.size()};
    bool ok = fwrite(f.type, 1, 4, fp) == 4 && fwrite(hdr, 4, 3, fp) == 3 &&
              fwrite(f.json.data(), 1, f.json.size(), fp) == f.json.size() &&
              fwrite(f.blob.data(), 1, f.blob.size(), fp) == f.blob.size();
    fclose(fp);
    if (!ok) err = "short write to " + path;
    return ok;
}

// ------------------------------------------------------------------ OBJ -----
// tinyobjloader's default behaviour as the baker uses it: all shapes merged,
// polygons fan-triangulated, 1-based or negative (relative) indices.
static bool load_obj(const std::string &path, std::vector<MeshVertex> &out, std::string &err) {
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) { err = "cannot open " + path; return false; }
    std::vector<float> pos, nor, tex;
    std::string text;                                  // one line of any length (a face of a large polygon does not fit a fixed buffer)
    for (;;) {
        text.clear();
        int ch;
        while ((ch = fgetc(fp)) != EOF && ch != '\n') text.push_back((char)ch);
        if (ch == EOF && text.empty()) break;
        text.push_back('\n');
        char *line = &text[0];
        char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float x = 0, y = 0, z = 0; sscanf(p + 2, "%f %f %f", &x, &y, &z);
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if (p[0] == 'v' && p[1] == 'n') {
            float x = 0, y = 0, z = 0; sscanf(p + 3, "%f %f %f", &x, &y, &z);
            nor.push_back(x); nor.push_back(y); nor.push_back(z);
        } else if (p[0] == 'v' && p[1] == 't') {
            float u = 0, v = 0; sscanf(p + 3, "%f %f", &u, &v);
            tex.push_back(u); tex.push_back(v);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<MeshVertex> corners;
            char *q = p + 2;
            while (*q) {
                while (*q == ' ' || *q == '\t') q++;
                if (*q == '\n' || *q == '\r' || *q == 0) break;
                long idx[3] = {0, 0, 0};   // v / vt / vn
                idx[0] = strtol(q, &q, 10);
                if (*q == '/') {
                    q++;
                    if (*q != '/' && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r' && *q) idx[1] = strtol(q, &q, 10);
                    if (*q == '/') { q++; idx[2] = strtol(q, &q, 10); }
                }
                while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;
                MeshVertex mv; memset(&mv, 0, sizeof mv);
                long nv = (long)pos.size() / 3, nt = (long)tex.size() / 2, nn = (long)nor.size() / 3;
                long vi = idx[0] > 0 ? idx[0] - 1 : nv + idx[0];
                long ti = idx[1] > 0 ? idx[1] - 1 : (idx[1] < 0 ? nt + idx[1] : -1);
                long ni = idx[2] > 0 ? idx[2] - 1 : (idx[2] < 0 ? nn + idx[2] : -1);
                if (vi < 0 || vi >= nv) { fclose(fp); err = "bad vertex index in " + path; return false; }
                memcpy(mv.pos, &pos[3 * vi], 12);
                if (ni >= 0 && ni < nn) memcpy(mv.normal, &nor[3 * ni], 12);
                float ux = 0, uy = 0;
                if (ti >= 0 && ti < nt) { ux = tex[2 * ti]; uy = tex[2 * ti + 1]; }
                mv.uv[0] = ux; mv.uv[1] = 1 - uy;                       // asset_main.cpp:47-48
                corners.push_back(mv);
            }
            for (size_t k = 2; k < corners.size(); k++) { out.push_back(corners[0]); out.push_back(corners[k - 1]); out.push_back(corners[k]); }
        }
    }
    fclose(fp);
    return true;
}

// A byte count read from an asset's JSON: finite, >= 0, a whole number of `elem`-byte elements, at most INT_MAX.
static bool valid_size(double v, size_t elem) {
    if (!(v >= 0.0) || !(v <= 2147483647.0) || v != std::floor(v)) return false;      // also rejects NaN
    return (size_t)v % elem == 0;
}

// mesh_asset.cpp:15-57 + vk_mesh.cpp:28-90
static bool load_mesh_container(const std::string &path, std::vector<MeshVertex> &verts, std::vector<uint32_t> &indices, std::string &err) {
    AssetFile f;
    if (!load_binaryfile(path, f, err)) return false;
    if (memcmp(f.type, "MESH", 4) != 0) { err = path + ": not a MESH asset"; return false; }
    double vbs = 0, ibs = 0; std::string fmt, comp;
    if (!json_number(f.json, "vertex_buffer_size", vbs) || !json_number(f.json, "index_buffer_size", ibs) ||
        !json_string(f.json, "vertex_format", fmt)) { err = path + ": missing mesh metadata"; return false; }
    if (fmt != "PNCV_F32") { err = path + ": unsupported vertex_format " + fmt; return false; }
    json_string(f.json, "compression", comp);
    // sizes come from the file: finite, non-negative, whole elements, and within what the int-based LZ4 codec can address
    if (!valid_size(vbs, sizeof(MeshVertex)) || !valid_size(ibs, 4) || vbs + ibs > 2147483647.0 || f.blob.size() > 2147483647u) {
        err = path + ": vertex_buffer_size / index_buffer_size out of range"; return false;
    }
    std::vector<uint8_t> raw((size_t)vbs + (size_t)ibs);
    if (comp == "LZ4") {
        if (lz4_decompress(f.blob.data(), (int)f.blob.size(), raw.data(), (int)raw.size()) != (int)raw.size()) { err = path + ": LZ4 stream corrupt"; return false; }
    } else {
        if (f.blob.size() != raw.size()) { err = path + ": blob size mismatch"; return false; }
        memcpy(raw.data(), f.blob.data(), raw.size());
    }
    verts.resize((size_t)vbs / sizeof(MeshVertex));
    memcpy(verts.data(), raw.data(), verts.size() * sizeof(MeshVertex));
    indices.resize((size_t)ibs / 4);
    memcpy(indices.data(), raw.data() + (size_t)vbs, indices.size() * 4);
    return true;
}

// Returns the de-indexed corner list (3 vertices per triangle).
bool load_mesh_file(const std::string &path, std::vector<MeshVertex> &verts, std::string &err) {
    if (path.size() > 4 && path.substr(path.size() - 4) == ".obj") return load_obj(path, verts, err);
    std::vector<MeshVertex> v; std::vector<uint32_t> idx;
    if (!load_mesh_container(path, v, idx, err)) return false;
    verts.clear(); verts.reserve(idx.size());
    for (uint32_t i : idx) { if (i >= v.size()) { err = path + ": index out of range"; return false; } verts.push_back(v[i]); }
    return true;
}

// texture_asset.cpp:46-70
bool load_texture_file(const std::string &path, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h, std::string &err) {
    AssetFile f;
    if (!load_binaryfile(path, f, err)) return false;
    if (memcmp(f.type, "TEXI", 4) != 0) { err = path + ": not a TEXI asset"; return false; }
    double ww = 0, hh = 0, bs = 0; std::string fmt, comp;
    if (!json_number(f.json, "width", ww) || !json_number(f.json, "height", hh) || !json_number(f.json, "buffer_size", bs) ||
        !json_string(f.json, "format", fmt)) { err = path + ": missing texture metadata"; return false; }
    if (!valid_size(ww, 1) || !valid_size(hh, 1) || !valid_size(bs, 4) || ww < 1.0 || hh < 1.0 || ww > 65536.0 || hh > 65536.0 || f.blob.size() > 2147483647u) {
        err = path + ": width / height / buffer_size out of range"; return false;
    }
    if (fmt != "RGBA8" || (size_t)bs != (size_t)ww * (size_t)hh * 4) { err = path + ": unsupported texture"; return false; }
    json_string(f.json, "compression", comp);
    rgba.resize((size_t)bs);
    if (comp == "LZ4") {
        if (lz4_decompress(f.blob.data(), (int)f.blob.size(), rgba.data(), (int)rgba.size()) != (int)rgba.size()) { err = path + ": LZ4 stream corrupt"; return false; }
    } else {
        if (f.blob.size() != rgba.size()) { err = path + ": blob size mismatch"; return false; }
        memcpy(rgba.data(), f.blob.data(), rgba.size());
    }
    w = (uint32_t)ww; h = (uint32_t)hh;
    return true;
}

}  // namespace crt

using namespace crt;

// Nothing may leave an extern "C" entry point as an exception: an allocation sized from a (validated, but possibly large)
// file, or any other failure inside the readers, becomes CRT_ERR_IO with a message.
template <typename F> static int guarded(F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { set_host_error("out of memory while reading an asset"); return CRT_ERR_IO; }
    catch (const std::exception &e) { set_host_error(e.what()); return CRT_ERR_IO; }
}

// malloc'ed copy handed to the caller (crt_free); never null on success, never a zero-length memcpy from a null source
template <typename T> static bool copy_out(const std::vector<T> &v, void **out) {
    void *p = malloc(v.size() * sizeof(T) + 1);
    if (!p) { set_host_error("out of memory while reading an asset"); return false; }
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    *out = p;
    return true;
}

extern "C" {

int crt_lz4_compress_bound(int n) { return lz4_bound(n); }
int crt_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap) { int r = lz4_compress(src, n, dst, cap); return r < 0 ? CRT_ERR_INVALID : r; }
int crt_lz4_decompress(const uint8_t *src, int n, uint8_t *dst, int cap) { int r = lz4_decompress(src, n, dst, cap); return r < 0 ? CRT_ERR_IO : r; }

int crt_obj_to_vertices(const char *path, float **vertices_pncv, uint32_t *n_vertices) {
    return guarded([&]() -> int {
    std::vector<MeshVertex> v; std::string err;
    if (!load_obj(path, v, err)) { set_host_error(err); return CRT_ERR_IO; }
    if (!copy_out(v, (void **)vertices_pncv)) return CRT_ERR_IO;
    *n_vertices = (uint32_t)v.size();
    return CRT_OK;
    });
}

int crt_load_obj(const char *path, uint32_t material_index, crt_tri **tris, uint32_t *n_tris) {
    return guarded([&]() -> int {
    std::vector<MeshVertex> v; std::string err;
    if (!load_mesh_file(path, v, err)) { set_host_error(err); return CRT_ERR_IO; }
    uint32_t n = (uint32_t)(v.size() / 3);
    crt_tri *t = (crt_tri *)calloc(n ? n : 1, sizeof(crt_tri));
    if (!t) { set_host_error("out of memory while reading an asset"); return CRT_ERR_IO; }
    for (uint32_t i = 0; i < n; i++) {
        const MeshVertex &a = v[3 * i], &b = v[3 * i + 1], &c = v[3 * i + 2];
        memcpy(t[i].v0, a.pos, 12); memcpy(t[i].v1, b.pos, 12); memcpy(t[i].v2, c.pos, 12);
        t[i].u[0] = a.uv[0]; t[i].u[1] = b.uv[0]; t[i].u[2] = c.uv[0];       // vk_engine.cpp:908-909
        t[i].v[0] = a.uv[1]; t[i].v[1] = b.uv[1]; t[i].v[2] = c.uv[1];
        t[i].materialIndex = material_index;
    }
    *tris = t; *n_tris = n;
    return CRT_OK;
    });
}

int crt_mesh_file_read(const char *path, float **vertices_pncv, uint32_t *n_vertices, uint32_t **indices, uint32_t *n_indices) {
    return guarded([&]() -> int {
    std::vector<MeshVertex> v; std::vector<uint32_t> idx; std::string err;
    if (!load_mesh_container(path, v, idx, err)) { set_host_error(err); return CRT_ERR_IO; }
    if (!copy_out(v, (void **)vertices_pncv)) return CRT_ERR_IO;
    if (!copy_out(idx, (void **)indices)) { free(*vertices_pncv); *vertices_pncv = nullptr; return CRT_ERR_IO; }
    *n_vertices = (uint32_t)v.size(); *n_indices = (uint32_t)idx.size();
    return CRT_OK;
    });
}

// mesh_asset.cpp:60-152 (pack_mesh + calculate_bounds)
int crt_mesh_file_write(const char *path, const float *vertices_pncv, uint32_t n_vertices, const uint32_t *indices,
                        uint32_t n_indices, const char *original_file) {
    return guarded([&]() -> int {
    const MeshVertex *v = (const MeshVertex *)vertices_pncv;
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {1.175494351e-38f, 1.175494351e-38f, 1.175494351e-38f};   // numeric_limits<float>::min() (sic, :118)
    for (uint32_t i = 0; i < n_vertices; i++) for (int a = 0; a < 3; a++) { mn[a] = std::fmin(mn[a], v[i].pos[a]); mx[a] = std::fmax(mx[a], v[i].pos[a]); }
    float ext[3], org[3];
    for (int a = 0; a < 3; a++) { ext[a] = (mx[a] - mn[a]) / 2.0f; org[a] = ext[a] + mn[a]; }
    float r2 = 0;
    for (uint32_t i = 0; i < n_vertices; i++) {
        float o0 = v[i].pos[0] - org[0], o1 = v[i].pos[1] - org[1], o2 = v[i].pos[2] - org[2];
        r2 = std::fmax(r2, o0 * o0 + o1 * o1 + o2 * o2);
    }
    AssetFile f; memcpy(f.type, "MESH", 4); f.version = 1;
    size_t vbs = (size_t)n_vertices * sizeof(MeshVertex), ibs = (size_t)n_indices * 4;
    char buf[512];
    snprintf(buf, sizeof buf, "{\"bounds\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g,%.9g],\"compression\":\"LZ4\",\"index_buffer_size\":%zu,\"index_size\":4,",
             org[0], org[1], org[2], std::sqrt(r2), ext[0], ext[1], ext[2], ibs);
    f.json = buf;
    f.json += "\"original_file\":\"" + json_escape(original_file ? original_file : "") + "\",";
    snprintf(buf, sizeof buf, "\"vertex_buffer_size\":%zu,\"vertex_format\":\"PNCV_F32\"}", vbs);
    f.json += buf;
    if (vbs + ibs > 2147483647u - 65536u) { set_host_error("mesh larger than the LZ4 block format addresses (2 GiB)"); return CRT_ERR_INVALID; }
    if ((vbs && !vertices_pncv) || (ibs && !indices)) { set_host_error("crt_mesh_file_write: a non-empty array has a null pointer"); return CRT_ERR_INVALID; }
    std::vector<uint8_t> merged(vbs + ibs);
    if (vbs) memcpy(merged.data(), vertices_pncv, vbs);
    if (ibs) memcpy(merged.data() + vbs, indices, ibs);
    f.blob.resize((size_t)lz4_bound((int)merged.size()));
    int c = lz4_compress(merged.data(), (int)merged.size(), f.blob.data(), (int)f.blob.size());
    if (c < 0) { set_host_error("LZ4 compression failed"); return CRT_ERR_INVALID; }
    f.blob.resize((size_t)c);
    std::string err;
    if (!save_binaryfile(path, f, err)) { set_host_error(err); return CRT_ERR_IO; }
    return CRT_OK;
    });
}

int crt_texture_file_read(const char *path, uint8_t **rgba8, uint32_t *width, uint32_t *height) {
    return guarded([&]() -> int {
    std::vector<uint8_t> px; std::string err;
    if (!load_texture_file(path, px, *width, *height, err)) { set_host_error(err); return CRT_ERR_IO; }
    if (!copy_out(px, (void **)rgba8)) return CRT_ERR_IO;
    return CRT_OK;
    });
}

// texture_asset.cpp:12-43
int crt_texture_file_write(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, const char *original_file) {
    return guarded([&]() -> int {
    AssetFile f; memcpy(f.type, "TEXI", 4); f.version = 1;
    size_t bs = (size_t)width * height * 4;
    char buf[256];
    snprintf(buf, sizeof buf, "{\"buffer_size\":%zu,\"compression\":\"LZ4\",\"format\":\"RGBA8\",\"height\":%u,", bs, height);
    f.json = buf;
    f.json += "\"original_file\":\"" + json_escape(original_file ? original_file : "") + "\",";
    snprintf(buf, sizeof buf, "\"width\":%u}", width);
    f.json += buf;
    if (bs > 2147483647u - 65536u) { set_host_error("texture larger than the LZ4 block format addresses (2 GiB)"); return CRT_ERR_INVALID; }
    f.blob.resize((size_t)lz4_bound((int)bs));
    int c = lz4_compress(rgba8, (int)bs, f.blob.data(), (int)f.blob.size());
    if (c < 0) { set_host_error("LZ4 compression failed"); return CRT_ERR_INVALID; }
    f.blob.resize((size_t)c);
    std::string err;
    if (!save_binaryfile(path, f, err)) { set_host_error(err); return CRT_ERR_IO; }
    return CRT_OK;
    });
}

}  // extern "C"
