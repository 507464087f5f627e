// Host-side matrix ingest for libspmv_hip: MatrixMarket coordinate files ->
// AoS COO, with the acceptance rules of the reference loader (reference
// src/mtx.c:36-137): banner "%%MatrixMarket matrix coordinate <type> <symmetry>",
// types real/double/integer/pattern (complex rejected), 1-based -> 0-based,
// explicit zeros dropped, pattern entries get 1.0, and every symmetry other than
// "general" mirrors off-diagonal entries with the same value (no sign flip for
// skew-symmetric, no conjugate for hermitian -- as the reference does).
//
// The whole file is read once; files laid out one entry per line (all the
// writers in use) are parsed by several threads, each taking a run of lines;
// anything else is tokenised sequentially (the reference calls fscanf per
// entry).  SPMV_HIP_IO_THREADS overrides the thread count (default: up to 16).
//
// Binary files use the layout of the reference's save_bin (src/mtx.c:207-247):
// int rows, int columns, long nnz, then nnz x {int i, int j, double a}, 0-based.

#include "spmv_hip.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace spmv { void set_error(const char *fmt, ...); void clear_error(); }
// sort.hip: (row, col) radix sort in HBM; 0 = sorted, non-zero = could not run (array untouched)
extern "C" int spmv_hip_internal_device_sort_coo(long long nnz, struct mtx *coo);

namespace {

struct Cursor {
    const char *p, *end;
    void skip_ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v')) p++; }
    bool token(std::string &out)
    {
        skip_ws();
        const char *b = p;
        while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
        out.assign(b, p);
        return p > b;
    }
    void rest_of_line(std::string &out)
    {
        const char *b = p;
        while (p < end && *p != '\n') p++;
        out.assign(b, p);
        if (p < end) p++;
    }
    bool integer(long &v)
    {
        skip_ws();
        if (p >= end) return false;
        char *e = nullptr;
        v = strtol(p, &e, 10);
        if (e == p) return false;
        p = e;
        return true;
    }
    bool real(double &v)
    {
        skip_ws();
        if (p >= end) return false;
        char *e = nullptr;
        v = strtod(p, &e);
        if (e == p) return false;
        p = e;
        return true;
    }
};

inline bool blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v'; }

// Parses the lines of [p, end): every non-blank line must be exactly "i j [value]".  Returns false when
// a line has another shape or a coordinate is out of range (the caller then re-parses sequentially, which
// also produces the error message).  `lines` counts entries read, `out` receives the kept ones in order.
bool parse_lines(const char *p, const char *end, long m, long n, bool pattern, bool mirrored,
                 std::vector<struct mtx> &out, long &lines_out)
{
    // local counter and output: neighbouring threads' slots share cache lines
    long lines = 0;
    std::vector<struct mtx> kept;
    kept.reserve((size_t)(end - p) / 12 * (mirrored ? 2 : 1));
    struct Publish {
        std::vector<struct mtx> &dst, &src; long &n_dst, &n_src;
        ~Publish() { dst.swap(src); n_dst = n_src; }
    } publish{out, kept, lines_out, lines};
    while (p < end) {
        while (p < end && blank(*p)) p++;
        if (p < end && *p == '\n') { p++; continue; }
        if (p >= end) break;
        char *e = nullptr;
        long i = strtol(p, &e, 10);
        if (e == p) return false;
        p = e;
        long j = strtol(p, &e, 10);
        if (e == p) return false;
        p = e;
        double v = 1.0;
        if (!pattern) {
            v = strtod(p, &e);
            if (e == p) return false;
            p = e;
        }
        while (p < end && blank(*p)) p++;
        if (p < end && *p != '\n') return false;              // more tokens on the line
        if (p < end) p++;
        lines++;
        if (!pattern && v == 0.0) continue;
        if (i < 1 || i > m || j < 1 || j > n) return false;
        kept.push_back({(int)i - 1, (int)j - 1, v});
        if (mirrored && i != j) kept.push_back({(int)j - 1, (int)i - 1, v});
    }
    return true;
}

int io_threads()
{
    const char *env = getenv("SPMV_HIP_IO_THREADS");
    int t = env ? atoi(env) : (int)std::min(std::thread::hardware_concurrency(), 16u);
    return std::max(t, 1);
}

// Splits the entry lines over threads at line ends.  Returns the number of entries kept, or -1 when the
// region is not "one entry per line, exactly nz lines" (nothing is written to coo in that case).
long parse_entries_parallel(const char *p, const char *end, long nz, long m, long n, bool pattern, bool mirrored,
                            struct mtx *coo)
{
    const size_t bytes = (size_t)(end - p);
    const int parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)io_threads(), bytes >> 20));
    std::vector<const char *> cut((size_t)parts + 1);
    cut[0] = p;
    cut[(size_t)parts] = end;
    for (int t = 1; t < parts; t++) {
        const char *q = std::max(p + bytes * (size_t)t / (size_t)parts, cut[(size_t)t - 1]);
        while (q < end && *q != '\n') q++;
        cut[(size_t)t] = q < end ? q + 1 : end;
    }
    std::vector<std::vector<struct mtx>> found((size_t)parts);
    std::vector<long> lines((size_t)parts, 0);
    std::vector<char> ok((size_t)parts, 0);
    auto work = [&](int t) {
        ok[(size_t)t] = parse_lines(cut[(size_t)t], cut[(size_t)t + 1], m, n, pattern, mirrored, found[(size_t)t],
                                    lines[(size_t)t]);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < parts; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    long total_lines = 0, kept = 0;
    for (int t = 0; t < parts; t++) {
        if (!ok[(size_t)t]) return -1;
        total_lines += lines[(size_t)t];
        kept += (long)found[(size_t)t].size();
    }
    if (total_lines != nz) return -1;
    long at = 0;
    for (int t = 0; t < parts; t++) {
        if (!found[(size_t)t].empty())
            memcpy(coo + at, found[(size_t)t].data(), sizeof(struct mtx) * found[(size_t)t].size());
        at += (long)found[(size_t)t].size();
    }
    return kept;
}

} // namespace

extern "C" {

void spmv_hip_sort_coo(int nnz, struct mtx *coo)
{
    auto less = [](const struct mtx &a, const struct mtx &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; };
    if (nnz < 2 || std::is_sorted(coo, coo + nnz, less)) return;
    // large arrays: radix sort in HBM (two PCIe passes of 16 B per entry instead of n log n compares);
    // the host sort remains for small arrays and when the device path cannot run
    const char *env = getenv("SPMV_HIP_DEVICE_SORT_MIN");
    const long threshold = env ? atol(env) : (1L << 20);
    if (nnz >= threshold && spmv_hip_internal_device_sort_coo(nnz, coo) == 0) return;
    std::stable_sort(coo, coo + nnz, less);
}

void spmv_hip_free_coo(struct mtx *coo) { free(coo); }

int spmv_hip_load_mtx(const char *fname, int *rows, int *columns, int *nnz, struct mtx **coo_out)
{
    spmv::clear_error();
    if (!fname || !rows || !columns || !nnz || !coo_out) { spmv::set_error("NULL argument"); return 1; }
    FILE *f = fopen(fname, "rb");
    if (!f) { spmv::set_error("Error opening file: %s", fname); return 1; }
    std::vector<char> buf;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz + 1 : 1);
        size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, f) : 0;
        buf[got] = 0;   // strtol/strtod stop here
        buf.resize(got + 1);
        fclose(f);
    }
    Cursor c{buf.data(), buf.data() + buf.size() - 1};

    std::string banner, object, format, type, symmetry;
    if (!c.token(banner) || !c.token(object) || !c.token(format) || !c.token(type) || !c.token(symmetry)) {
        spmv::set_error("Error reading header from file: %s", fname);
        return 1;
    }
    if (banner != "%%MatrixMarket") { spmv::set_error("Missing header in file: %s", fname); return 1; }
    if (object != "matrix" || format != "coordinate") {
        spmv::set_error("Object (%s) and/or format (%s) not supported in file: %s", object.c_str(), format.c_str(), fname);
        return 1;
    }
    // comment lines start with '%' in column 0; the first other line holds the sizes
    c.skip_ws();
    std::string line;
    for (;;) {
        if (c.p >= c.end) { spmv::set_error("Error reading matrix size from file: %s", fname); return 1; }
        if (*c.p != '%') break;
        c.rest_of_line(line);
    }
    long m, n, nz;
    if (!c.integer(m) || !c.integer(n) || !c.integer(nz)) {
        spmv::set_error("Error reading matrix size from file: %s", fname);
        return 1;
    }
    if (m > INT_MAX || n > INT_MAX || nz > INT_MAX || m < 0 || n < 0 || nz < 0) {
        spmv::set_error("Matrix too big (%ld %ld %ld): %s", m, n, nz, fname);
        return 1;
    }
    if (type == "complex") { spmv::set_error("Matrix is complex (%s): %s", type.c_str(), fname); return 1; }
    const bool mirrored = symmetry != "general";
    const bool pattern = type == "pattern";
    if (mirrored && 2 * nz > INT_MAX) { spmv::set_error("Matrix too big once mirrored: %s", fname); return 1; }

    struct mtx *coo = (struct mtx *)malloc(sizeof(struct mtx) * (size_t)std::max<long>(mirrored ? 2 * nz : nz, 1));
    if (!coo) { spmv::set_error("Out of memory loading: %s", fname); return 1; }
    long k = parse_entries_parallel(c.p, c.end, nz, m, n, pattern, mirrored, coo);
    if (k >= 0) {
        *rows = (int)m; *columns = (int)n; *nnz = (int)k; *coo_out = coo;
        return 0;
    }
    // irregular layout (entries not one per line, or more lines than entries): token by token, as fscanf would
    k = 0;
    for (long l = 0; l < nz; l++) {
        long i, j;
        double v = 1.0;
        if (!c.integer(i) || !c.integer(j) || (!pattern && !c.real(v))) {
            spmv::set_error("Error reading matrix element %ld", l);
            free(coo);
            return 1;
        }
        if (!pattern && v == 0.0) continue;
        if (i < 1 || i > m || j < 1 || j > n) {
            spmv::set_error("Coordinates (%ld,%ld) out of range in line %ld", i, j, l);
            free(coo);
            return 1;
        }
        coo[k].i = (int)i - 1; coo[k].j = (int)j - 1; coo[k].a = v; k++;
        if (mirrored && i != j) { coo[k].i = (int)j - 1; coo[k].j = (int)i - 1; coo[k].a = v; k++; }
    }
    *rows = (int)m; *columns = (int)n; *nnz = (int)k; *coo_out = coo;
    return 0;
}

int spmv_hip_load_bin(const char *fname, int *rows, int *columns, int *nnz, struct mtx **coo_out)
{
    spmv::clear_error();
    if (!fname || !rows || !columns || !nnz || !coo_out) { spmv::set_error("NULL argument"); return 1; }
    static_assert(sizeof(struct mtx) == 16, "struct mtx is {int, int, double} without padding");
    FILE *f = fopen(fname, "rb");
    if (!f) { spmv::set_error("Error opening file: %s", fname); return 1; }
    int m = 0, n = 0;
    long nz = 0;
    if (fread(&m, sizeof m, 1, f) != 1 || fread(&n, sizeof n, 1, f) != 1 || fread(&nz, sizeof nz, 1, f) != 1) {
        spmv::set_error("Error reading file: %s", fname);
        fclose(f);
        return 1;
    }
    if (m < 0 || n < 0 || nz < 0 || nz > INT_MAX) {
        spmv::set_error("Matrix too big (%d %d %ld): %s", m, n, nz, fname);
        fclose(f);
        return 1;
    }
    struct mtx *coo = (struct mtx *)malloc(sizeof(struct mtx) * (size_t)std::max<long>(nz, 1));
    if (!coo) { spmv::set_error("Out of memory loading: %s", fname); fclose(f); return 1; }
    if (fread(coo, sizeof(struct mtx), (size_t)nz, f) != (size_t)nz) {
        spmv::set_error("Error reading file: %s", fname);
        free(coo);
        fclose(f);
        return 1;
    }
    fclose(f);
    for (long k = 0; k < nz; k++)
        if (coo[k].i < 0 || coo[k].i >= m || coo[k].j < 0 || coo[k].j >= n) {
            spmv::set_error("Coordinates (%d,%d) out of range in entry %ld of %s", coo[k].i, coo[k].j, k, fname);
            free(coo);
            return 1;
        }
    *rows = m; *columns = n; *nnz = (int)nz; *coo_out = coo;
    return 0;
}

int spmv_hip_save_bin(const char *fname, int rows, int columns, int nnz, const struct mtx *coo)
{
    spmv::clear_error();
    if (!fname || (nnz > 0 && !coo) || rows < 0 || columns < 0 || nnz < 0) { spmv::set_error("bad argument"); return 1; }
    FILE *f = fopen(fname, "wb");
    if (!f) { spmv::set_error("Error opening file: %s", fname); return 1; }
    const long nz = nnz;
    const bool ok = fwrite(&rows, sizeof rows, 1, f) == 1 && fwrite(&columns, sizeof columns, 1, f) == 1 &&
                    fwrite(&nz, sizeof nz, 1, f) == 1 && fwrite(coo, sizeof(struct mtx), (size_t)nnz, f) == (size_t)nnz;
    if (fclose(f) != 0 || !ok) { spmv::set_error("Error writing file: %s", fname); return 1; }
    return 0;
}

} // extern "C"
