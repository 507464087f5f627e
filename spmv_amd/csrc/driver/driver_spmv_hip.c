/*
 * spmv_hip -- C driver for libspmv_hip with the command line, selection logic,
 * measurement protocol and CSV schema of the reference driver
 * (reference src/driver_spmv.c:112-347), so that it drops in behind spmv.run:
 *
 *     build/spmv_hip MODE LIST OUT.csv
 *
 * MODE is one of the reference's mode strings (driver_spmv.c:186-261) -- each
 * selects the same storage format, now resident in HBM -- optionally followed by
 * ":"-separated suffixes: "f32" / "f64" (storage and accumulation type) and, for
 * sell / sell_dfc, a number = the sorting window sigma ("sell:1024", "sell:1024:f32").
 * The reference keeps sigma and C in two globals (sell.c:4-5: 16384 and 64); without
 * a suffix the environment variable SPMV_SIGMA is consulted, then that default.
 * SPMV_EXACT_ORDER=0 lets the kernel trial also pick variants that reassociate the
 * row sums (default: only variants whose y has the same bits run to run).
 * SPMV_NGPUS=n (n > 1) cuts every matrix into n row blocks of equal nnz shares, one
 * per GPU, driven from this one thread (spmv_hip_dist_*; the library issues the blocks'
 * launches from a hidden helper thread per GPU); SPMV_DEVICES="0,1,.." names the
 * devices (default 0..n-1; a device may repeat).  The CSV carries the count in its
 * NGPUs column; one line after the table header says which pairs of GPUs exchange
 * their halo directly and which through host memory (SPMV_REQUIRE_P2P=1 refuses the
 * latter).  MODE auto picks a format per matrix on one GPU and is refused with n > 1.
 * Per matrix of LIST the driver
 *   1. estimates the memory need like enought_memory (driver_spmv.c:50-110;
 *      .bin headers are read as binary here -- the reference parses them as text),
 *   2. loads it (.mtx: spmv_hip_load_mtx; .bin: spmv_hip_load_bin, the reference's
 *      save_bin layout, mtx.c:207-247) and insists on a square matrix (driver_spmv.c:176-179),
 *   3. builds the format on the device,
 *   4. fills x, y with the reference's interleaved rand() sequence and
 *      normalises x (driver_spmv.c:267-273),
 *   5. checks one product against the host COO loop (mult_mtx, mtx.c:270-276)
 *      with the relative L2 error and the 1e-10 bar (driver_spmv.c:274-289,313),
 *   6. runs the timed power loop with x, y resident in HBM (driver_spmv.c:292-305
 *      -- only the product is timed, >= 10 iterations and > 2 s),
 *   7. prints the table row and the CSV row (driver_spmv.c:311-316) with extra
 *      columns appended: GB/s (algorithmic bytes), % of 8 TB/s, dtype, kernel variant.
 * Library errors are printed and end the run with exit code 1, as the reference's
 * helpers do (mtx.c:39-41, spmv.h:277-281).
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_hip.h"

#define MAX_PATH_LEN 1024
#define REFERENCE_BLOCK 4            /* _BLOCK of the reference's AVX2 build (spmv.h:20-22) */
#define HBM_PEAK_GBS 8000.0

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, spmv_hip_last_error());
    exit(1);
}

static double vec_norm(int n, const double *x)            /* driver_spmv.c:21-26 */
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return sqrt(s);
}

static unsigned long available_memory_kb(void)            /* driver_spmv.c:29-48 */
{
    FILE *f = fopen("/proc/meminfo", "r");
    if (!f) return 0;
    char line[256];
    unsigned long kb = 0;
    while (fgets(line, sizeof line, f)) {
        if (strncmp(line, "MemAvailable:", 13) == 0) {
            char *p = line + 13;
            while (*p && !isdigit((unsigned char)*p)) p++;
            kb = strtoul(p, NULL, 10);
            break;
        }
    }
    fclose(f);
    return kb;
}

static int has_suffix(const char *s, const char *suffix)
{
    size_t a = strlen(s), b = strlen(suffix);
    return a >= b && strcmp(s + a - b, suffix) == 0;
}

static int g_ngpus = 1;
static int g_devices[64];

/* Free HBM over the distinct devices of the run, in MB (0 = unknown). */
static double free_hbm_mb(void)
{
    double total = 0.0;
    for (int p = 0; p < g_ngpus; p++) {
        int seen = 0;
        for (int q = 0; q < p; q++) seen |= g_devices[q] == g_devices[p];
        if (seen) continue;
        size_t hbm_free = 0;
        if (spmv_hip_init(g_devices[p]) || spmv_hip_mem_info(&hbm_free, NULL)) { total = 0.0; break; }
        total += (double)(hbm_free / (1024 * 1024));
    }
    (void)spmv_hip_init(g_devices[0]);
    return total;
}

/* Sizes from the file header, and the reference's estimate 2*16*nnz + 8*(rows+cols) bytes in MB. */
static int enough_memory(const char *path, long *rows, long *cols, long *nnz, unsigned long *need_mb)
{
    *rows = *cols = *nnz = 0;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 0; }
    if (has_suffix(path, ".bin")) {
        int m, n; long nz;
        if (fread(&m, sizeof m, 1, f) != 1 || fread(&n, sizeof n, 1, f) != 1 || fread(&nz, sizeof nz, 1, f) != 1) { fclose(f); return 0; }
        *rows = m; *cols = n; *nnz = nz;
    } else {
        char line[1024];
        int coordinate = 0;
        if (fgets(line, sizeof line, f) && strstr(line, "coordinate")) coordinate = 1;
        while (fgets(line, sizeof line, f)) if (line[0] != '%') break;
        if (coordinate) sscanf(line, "%ld %ld %ld", rows, cols, nnz);
        else { sscanf(line, "%ld %ld", rows, cols); *nnz = *rows * *cols; }
    }
    fclose(f);
    unsigned long bytes = 2ul * (unsigned long)*nnz * 16ul + (unsigned long)(*rows + *cols) * 8ul;
    *need_mb = bytes / (1024 * 1024);
    unsigned long free_mb = available_memory_kb() / 1024;
    if (!(*need_mb < free_mb * 0.95)) return 0;
    /* the device holds the CSR arrays and the format built from them at the same time: about the same
     * 2 * 16 bytes per entry, against what is free in HBM (all GPUs of a row-block run together) */
    const double hbm_mb = free_hbm_mb();
    if (hbm_mb > 0.0) return (double)*need_mb < hbm_mb * 0.95;
    return 1;
}

/* COO -> plain CSR arrays on the host, for the creators that take CSR (the reference driver gets them from create_csr, driver_spmv.c:207,243) */
static void coo_to_csr(int rows, int nnz, const struct mtx *coo, int **rp, int **cj, double **va)
{
    *rp = (int *)calloc((size_t)rows + 1, sizeof(int));
    *cj = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    *va = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    for (int l = 0; l < nnz; l++) (*rp)[coo[l].i + 1]++;
    for (int r = 0; r < rows; r++) (*rp)[r + 1] += (*rp)[r];
    for (int l = 0; l < nnz; l++) { (*cj)[l] = coo[l].j; (*va)[l] = coo[l].a; }
}

/* MODE -> handle: the if/else chain of driver_spmv.c:186-261 */
static spmv_hip_matrix *create_for_mode(const char *mode, int rows, int cols, int nnz, struct mtx *coo, int dtype,
                                        long sigma)
{
    /* "coo" (driver_spmv.c:186-188): create_coo sorts by (row, col) and mult_coo (coo.c:3-14) sums each
     * row's entries in that order from 0.0 -- the sums of mult_csr_base, so it runs on the lane-per-row
     * CSR kernel */
    if (!strcmp(mode, "coo") || !strcmp(mode, "csr_base") || !strcmp(mode, "csr_autovec"))
        return spmv_hip_create_csr(rows, cols, nnz, 1, coo, dtype, SPMV_HIP_CSR_SCALAR, 0);
    if (!strcmp(mode, "csr_vec") || !strcmp(mode, "csr_highway"))
        return spmv_hip_create_csr(rows, cols, nnz, REFERENCE_BLOCK, coo, dtype, SPMV_HIP_CSR_VECTOR, 0);
    if (!strcmp(mode, "csr_merge")) return spmv_hip_create_csr_merge(rows, cols, nnz, coo, dtype, 0);
    if (!strcmp(mode, "csr_bal")) return spmv_hip_create_csr_bal(rows, cols, nnz, coo, dtype, 0);
    if (!strcmp(mode, "ell")) return spmv_hip_create_ell(rows, cols, nnz, coo, dtype, 0);
    if (!strcmp(mode, "auto")) return spmv_hip_create_auto(rows, cols, nnz, coo, dtype, 3);   /* format by trial */
    if (!strcmp(mode, "sellp_vec") || !strcmp(mode, "sellp_autovec") || !strcmp(mode, "sellp_highway") || !strcmp(mode, "sell") ||
        !strcmp(mode, "sell_dfc")) {
        int *rp, *cj; double *va;
        spmv_hip_sort_coo(nnz, coo);                       /* create_csr's side effect (csr.c:167) */
        coo_to_csr(rows, nnz, coo, &rp, &cj, &va);
        spmv_hip_matrix *m = !strcmp(mode, "sell")       ? spmv_hip_create_sell(rows, cols, nnz, rp, cj, va, dtype, 0, sigma)
                             : !strcmp(mode, "sell_dfc") ? spmv_hip_create_sell_dfc(rows, cols, nnz, rp, cj, va, dtype, 0, sigma)
                                                         : spmv_hip_create_sellp(rows, cols, nnz, rp, cj, va, dtype, 0);
        free(rp); free(cj); free(va);
        return m;
    }
    printf("Error: Unknown execution mode.\n");            /* driver_spmv.c:258-260 */
    exit(-1);
}

/* MODE -> storage format of the interior matrices of a row-block run (the MODE chain above without
 * the per-mode host conversions: every block is built from its CSR in HBM) */
static int format_for_mode(const char *mode, long sigma, long *p0, long *p1)
{
    *p0 = *p1 = 0;
    if (!strcmp(mode, "coo") || !strcmp(mode, "csr_base") || !strcmp(mode, "csr_autovec")) return SPMV_HIP_CSR_SCALAR;
    if (!strcmp(mode, "csr_vec") || !strcmp(mode, "csr_highway")) return SPMV_HIP_CSR_VECTOR;
    if (!strcmp(mode, "csr_merge")) return SPMV_HIP_CSR_MERGE;
    if (!strcmp(mode, "csr_bal")) return SPMV_HIP_CSR_BAL;
    if (!strcmp(mode, "ell")) return SPMV_HIP_ELL;
    if (!strcmp(mode, "sellp_vec") || !strcmp(mode, "sellp_autovec") || !strcmp(mode, "sellp_highway")) return SPMV_HIP_SELLP;
    if (!strcmp(mode, "sell")) { *p0 = 64; *p1 = sigma; return SPMV_HIP_SELL; }
    if (!strcmp(mode, "sell_dfc")) { *p0 = 64; *p1 = sigma; return SPMV_HIP_SELL_DFC; }
    printf("Error: Unknown execution mode.\n");            /* driver_spmv.c:258-260 */
    exit(-1);
}

int main(int argc, char *argv[])
{
    if (argc != 4) {
        printf("Usage: spmv_hip [coo|csr_base|csr_vec|csr_merge|csr_bal|sellp_vec|ell|sell|sell_dfc|auto][:sigma][:f32] <mtx list> <output csv>\n");
        exit(-1);
    }
    char mode[64];
    snprintf(mode, sizeof mode, "%s", argv[1]);
    int dtype = SPMV_HIP_F64;
    long sigma = getenv("SPMV_SIGMA") ? atol(getenv("SPMV_SIGMA")) : 0;       /* 0 = the reference's 16384 (sell.c:4) */
    char *colon = strchr(mode, ':');
    if (colon) *colon++ = 0;
    while (colon && *colon) {                                                /* suffixes: f32 | f64 | <sigma> */
        char *next = strchr(colon, ':');
        if (next) *next++ = 0;
        char *end;
        long number = strtol(colon, &end, 10);
        if (!strcmp(colon, "f32")) dtype = SPMV_HIP_F32;
        else if (!strcmp(colon, "f64")) dtype = SPMV_HIP_F64;
        else if (end != colon && !*end && number > 0 && (!strcmp(mode, "sell") || !strcmp(mode, "sell_dfc"))) sigma = number;
        else { printf("Error: Unknown execution mode.\n"); exit(-1); }
        colon = next;
    }
    const char *exact_env = getenv("SPMV_EXACT_ORDER");
    const int exact_order = exact_env ? atoi(exact_env) != 0 : 1;
    int ngpus = getenv("SPMV_NGPUS") ? atoi(getenv("SPMV_NGPUS")) : 1;
    if (ngpus < 1 || ngpus > 64) { printf("Error: SPMV_NGPUS must be in 1..64\n"); exit(-1); }
    g_ngpus = ngpus;
    int *devices = g_devices;
    for (int p = 0; p < 64; p++) devices[p] = ngpus > 1 ? p : 0;
    if (getenv("SPMV_DEVICES")) {
        char list[256];
        snprintf(list, sizeof list, "%s", getenv("SPMV_DEVICES"));
        int p = 0;
        for (char *tok = strtok(list, ","); tok && p < 64; tok = strtok(NULL, ",")) devices[p++] = atoi(tok);
        if (p < ngpus) { printf("Error: SPMV_DEVICES names %d devices, SPMV_NGPUS is %d\n", p, ngpus); exit(-1); }
    }
    const int vbytes = dtype == SPMV_HIP_F32 ? 4 : 8;
    if (ngpus > 1) {                                       /* before any file is opened or written */
        if (!strcmp(mode, "auto")) {
            printf("Error: MODE auto picks a format per matrix on one GPU; with SPMV_NGPUS > 1 name the format.\n");
            exit(-1);
        }
        long p0, p1;
        (void)format_for_mode(mode, sigma, &p0, &p1);      /* exits on an unknown MODE */
    }
    FILE *infd = fopen(argv[2], "r");
    if (!infd) { perror("Error openning file input for matrix list.\n"); exit(-1); }
    FILE *outfd = fopen(argv[3], "w");
    if (!outfd) { perror(argv[3]); exit(-1); }
    if (spmv_hip_init(devices[0])) die("spmv_hip_init");

    fprintf(outfd, "Format;File;Rows;Columns;NNZ;GFlops;Time(s);GB/s;PctHBMPeak;Dtype;Variant;NGPUs\n");
    printf("\n  Format: %s   dtype: %s   GPUs: %d   list: %s   csv: %s\n\n", mode, dtype == SPMV_HIP_F32 ? "f32" : "f64", ngpus, argv[2], argv[3]);
    printf("  %-30s %10s %10s %11s | %10s %11s %9s | %8s %6s | %11s  %s\n", "Matrix", "ROWS", "COLUMNS", "NNZ", "GFlops",
           "T(s)", "Mem (Mb)", "GB/s", "%HBM", "ERROR", "Test");

    char line[MAX_PATH_LEN];
    int peers_printed = 0;
    while (fgets(line, sizeof line, infd)) {
        line[strcspn(line, "\r\n")] = 0;
        if (!line[0]) continue;
        const char *slash = strrchr(line, '/');
        const char *name = slash ? slash + 1 : line;
        long hrows, hcols, hnnz;
        unsigned long need_mb;
        if (!enough_memory(line, &hrows, &hcols, &hnnz, &need_mb)) {      /* driver_spmv.c:332-337 */
            printf("  %-30s %10ld %10ld %11ld | %10.4f %11.4e %9lu | %8.1f %6.1f | %11.4e  ---\n", name, hrows, hcols, hnnz,
                   0.0, 0.0, need_mb, 0.0, 0.0, 0.0);
            fprintf(outfd, "%s;%s;%ld;%ld;%ld;%.5e;%.5e;%.5e;%.2f;%s;%d;%d\n", argv[1], name, hrows, hcols, hnnz, 0.0, 0.0, 0.0, 0.0,
                    dtype == SPMV_HIP_F32 ? "f32" : "f64", 0, ngpus);
            continue;
        }
        int rows, columns, nnz;
        struct mtx *coo;
        if (has_suffix(line, ".bin")) { if (spmv_hip_load_bin(line, &rows, &columns, &nnz, &coo)) die("load_bin"); }
        else if (has_suffix(line, ".mtx")) { if (spmv_hip_load_mtx(line, &rows, &columns, &nnz, &coo)) die("load_mtx"); }
        else { printf("Unkown file extension\n"); return 1; }
        if (rows != columns) { printf("Matrix is not square\n"); return 1; }

        spmv_hip_matrix *m = NULL;
        spmv_hip_dist *dist = NULL;
        if (ngpus > 1) {
            long p0, p1;
            const int format = format_for_mode(mode, sigma, &p0, &p1);
            dist = spmv_hip_dist_create(format, rows, columns, nnz, coo, dtype, p0, p1, ngpus, devices);
            if (!dist) die("dist_create");
            /* a new handle runs its default kernels: restrict the trial first, then tune */
            if (spmv_hip_dist_set_option(dist, "exact_order", exact_order)) die("dist_set_option");
            if (spmv_hip_dist_analyze(dist, 3)) die("dist_analyze");
            if (!peers_printed) {
                int access[64 * 64], staged = 0, direct = 0;
                if (spmv_hip_dist_get_peer_access(dist, access)) die("dist_get_peer_access");
                for (int q = 0; q < ngpus; q++)
                    for (int p = 0; p < ngpus; p++)
                        if (p != q) { staged += access[q * ngpus + p] == 0; direct += access[q * ngpus + p] == 1; }
                printf("  halo exchange: %d pairs peer-to-peer, %d pairs through host memory, %d pairs on one device\n",
                       direct, staged, ngpus * (ngpus - 1) - direct - staged);
                peers_printed = 1;
            }
        } else {
            m = create_for_mode(mode, rows, columns, nnz, coo, dtype, sigma);
            if (!m) die("create");
            if (spmv_hip_set_option(m, "exact_order", exact_order)) die("set_option");
            if (spmv_hip_analyze(m, 3)) die("analyze");
        }

        double *x = (double *)malloc(sizeof(double) * (size_t)columns);
        double *y = (double *)malloc(sizeof(double) * ((size_t)rows + 64));
        double *y_ref = (double *)malloc(sizeof(double) * (size_t)rows);
        for (int i = 0; i < columns; i++) {                               /* driver_spmv.c:267-270 */
            x[i] = ((double)rand()) / RAND_MAX;
            y[i] = ((double)rand()) / RAND_MAX;
        }
        double norm = vec_norm(columns, x);
        for (int i = 0; i < columns; i++) x[i] /= norm;
        if (dtype == SPMV_HIP_F32)                     /* compare against what the device will see */
            for (int i = 0; i < columns; i++) x[i] = (double)(float)x[i];
        for (int i = 0; i < rows; i++) y_ref[i] = 0.0;                    /* mult_mtx, mtx.c:270-276 */
        for (int k = 0; k < nnz; k++) y_ref[coo[k].i] += x[coo[k].j] * coo[k].a;

        if (dist ? spmv_hip_dist_mult(dist, x, y) : spmv_hip_mult(m, x, y)) die("mult");
        double error = 0.0, nrm = 0.0;                                    /* driver_spmv.c:278-289 */
        for (int i = 0; i < rows; i++) {
            nrm += y_ref[i] * y_ref[i];
            double d = fabs(y[i] - y_ref[i]);
            error += d * d;
        }
        error = nrm != 0.0 ? sqrt(error) / sqrt(nrm) : sqrt(error);

        int it = 0;
        double time_spmv = 0.0;
        const char *ms = getenv("SPMV_MIN_SECONDS");                      /* default: the reference's 2 s (:303) */
        if (dist ? spmv_hip_dist_bench(dist, 10, ms ? atof(ms) : 2.0, x, &it, &time_spmv)
                 : spmv_hip_bench(m, 10, ms ? atof(ms) : 2.0, x, &it, &time_spmv)) die("bench");
        double gflops = 2.0 * nnz * it / time_spmv / 1e9;                 /* driver_spmv.c:307-308 */
        double ftime = time_spmv / it;
        double bytes = (double)nnz * (vbytes + 4) + (double)rows * (4 + vbytes) + (double)columns * vbytes;
        double gbs = bytes / ftime / 1e9;
        spmv_hip_info info;
        memset(&info, 0, sizeof info);
        if (m) spmv_hip_get_info(m, &info);
        else {                                             /* the interior kernel variant of block 0 */
            spmv_hip_rowblock_info blocks[64];
            if (spmv_hip_dist_get_info(dist, NULL, NULL, blocks) == 0) info.variant = blocks[0].interior_variant;
        }
        const double bar = dtype == SPMV_HIP_F32 ? 1e-5 : 1e-10;          /* driver_spmv.c:313 (fp64) */
        const double peak = HBM_PEAK_GBS * ngpus;          /* the roofline of all GPUs taking part */
        printf("  %-30s %10d %10d %11d | %10.4f %11.4e %9lu | %8.1f %6.1f | %11.4e  %s\n", name, rows, columns, nnz, gflops,
               ftime, need_mb, gbs, 100.0 * gbs / peak, error, error < bar ? "OK" : "ERR");
        fprintf(outfd, "%s;%s;%d;%d;%d;%.5e;%.5e;%.5e;%.2f;%s;%d;%d\n", argv[1], name, rows, columns, nnz, gflops, ftime, gbs,
                100.0 * gbs / peak, dtype == SPMV_HIP_F32 ? "f32" : "f64", info.variant, ngpus);
        fflush(outfd);

        free(x); free(y); free(y_ref);
        spmv_hip_free_coo(coo);
        if (m) spmv_hip_free(m);
        if (dist) spmv_hip_dist_free(dist);
    }
    printf("\n");
    fclose(infd);
    fclose(outfd);
    return 0;
}
