// Host-side matrix ingest for libspmv_hip: MatrixMarket coordinate files ->
// AoS COO, with the acceptance rules of the reference loader (reference
// src/mtx.c:36-137): banner "%%MatrixMarket matrix coordinate <type> <symmetry>",
// types real/double/integer/pattern (complex rejected), 1-based -> 0-based,
// explicit zeros dropped, pattern entries get 1.0, and every symmetry other than
// "general" mirrors off-diagonal entries with the same value (no sign flip for
// skew-symmetric, no conjugate for hermitian -- as the reference does).
//
// The whole file is read once and tokenised in memory (the reference calls
// fscanf per entry).

#include "spmv_hip.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace spmv { void set_error(const char *fmt, ...); void clear_error(); }

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

} // namespace

extern "C" {

void spmv_hip_sort_coo(int nnz, struct mtx *coo)
{
    auto less = [](const struct mtx &a, const struct mtx &b) { return a.i != b.i ? a.i < b.i : a.j < b.j; };
    if (nnz > 1 && !std::is_sorted(coo, coo + nnz, less)) std::sort(coo, coo + nnz, less);
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
    long k = 0;
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

} // extern "C"
