// Host-side check of fluidlab_amd/csrc/fe_math.h (the math the HIP kernels inline).
// Built twice by tests/test_csrc_math.py: -DFE_T=double (finite-difference check of the
// hand-derived adjoints) and -DFE_T=float (SVD contract at the product precision).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../fluidlab_amd/csrc/fe_math.h"

static double urand() { return rand() / (double)RAND_MAX; }
static double nrand() { double s = 0; for (int i = 0; i < 12; i++) s += urand(); return s - 6.0; }
static m3 rand_m3(double scale) { m3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = (real)(scale * nrand()); return r; }

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static double svd_residual(const m3& F, double* orth, double* dets) {
    m3 U, V; real sig[3];
    svd3(F, U, sig, V);
    m3 US = U; for (int d = 0; d < 3; d++) for (int i = 0; i < 3; i++) US.a[i][d] *= sig[d];
    m3 R = m3_mul_nt(US, V);
    double res = 0, nrm = 1e-30;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { res = fmax(res, fabs((double)R.a[i][j] - F.a[i][j])); nrm = fmax(nrm, fabs((double)F.a[i][j])); }
    m3 UtU = m3_mul_tn(U, U), VtV = m3_mul_tn(V, V);
    double o = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { o = fmax(o, fabs((double)UtU.a[i][j] - (i == j))); o = fmax(o, fabs((double)VtV.a[i][j] - (i == j))); }
    *orth = o;
    *dets = fmin((double)m3_det(U), (double)m3_det(V));
    CHECK(fabs(sig[0]) >= fabs(sig[1]) - 1e-6 * fabs(sig[0]) && fabs(sig[1]) >= fabs(sig[2]) - 1e-6 * fabs(sig[0]), "sigma not descending %g %g %g", (double)sig[0], (double)sig[1], (double)sig[2]);
    CHECK(sig[0] >= 0 && sig[1] >= 0, "only the last sigma may be negative");
    CHECK((m3_det(F) < 0) == (sig[2] < 0) || fabs((double)sig[2]) < 1e-6 * fabs((double)sig[0]), "sign of sigma[2] must follow det F");
    return res / nrm;
}

static double loss_of(const m3& C, const m3& F, double dt, double mu, double lam, double mass, int cls, double scale, const m3& wA, const m3& wF) {
    Constitutive k;
    constitutive_eval(C, F, (real)dt, (real)mu, (real)lam, (real)mass, cls, (real)scale, k);
    double L = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L += (double)wA.a[i][j] * k.affine.a[i][j] + (double)wF.a[i][j] * k.Fnew.a[i][j];
    return L;
}

int main() {
    srand(7);
    const double tol_res = sizeof(real) == 4 ? 3e-6 : 1e-7, tol_orth = sizeof(real) == 4 ? 3e-6 : 1e-7;
    // ---- SVD contract
    for (int t = 0; t < 2000; t++) {
        m3 F;
        int kind = t % 5;
        if (kind == 0) F = m3_add(m3_ident(), rand_m3(0.02));           // fluid-like
        else if (kind == 1) F = rand_m3(1.0);                            // generic (half have det < 0)
        else if (kind == 2) { F = rand_m3(1.0); for (int i = 0; i < 3; i++) F.a[i][2] = F.a[i][0] * (real)0.5 - F.a[i][1]; }   // rank 2
        else if (kind == 3) { m3 a = rand_m3(1.0); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F.a[i][j] = a.a[i][0] * a.a[j][1]; }  // rank 1
        else F = m3_scale(m3_ident(), (real)(0.5 + urand()));            // all sigma equal
        double orth, dets;
        double res = svd_residual(F, &orth, &dets);
        CHECK(res < tol_res, "svd residual %g (kind %d)", res, kind);
        CHECK(orth < tol_orth, "svd orthogonality %g (kind %d)", orth, kind);
        CHECK(dets > 0.99, "U, V must be proper rotations (kind %d)", kind);
    }
    { m3 Z = m3_zero(); double orth, dets; double res = svd_residual(Z, &orth, &dets); CHECK(res < 1e-6 && orth < 1e-6 && dets > 0.99, "zero matrix"); }
    // ---- adjoint of the constitutive model vs central finite differences (fp64 build only)
    if (sizeof(real) == 8) {
        struct { const char* name; int cls; double mu; double fnoise; } cases[] = {
            {"liquid mu=0 (no-SVD path)", FE_MAT_LIQUID_, 0.0, 0.05}, {"liquid mu=200", FE_MAT_LIQUID_, 200.0, 0.05},
            {"elastic", FE_MAT_ELASTIC_, 416.67, 0.05}, {"plasto-elastic", FE_MAT_PLASTO_ELASTIC_, 416.67, 0.002},
            {"plasto-elastic demo", FE_MAT_PLASTO_ELASTIC_DEMO_, 160.0, 0.002}};
        const double dt = 2e-4, lam = 277.78, mass = 6.1e-5, n = 64, p_vol = (0.5 / n) * (0.5 / n), scale = -dt * p_vol * 4 * n * n;
        for (auto& cs : cases) {
            double worst = 0;
            for (int t = 0; t < 20; t++) {
                m3 C = rand_m3(5.0), F = m3_add(m3_ident(), rand_m3(cs.fnoise)), wA = rand_m3(1.0), wF = rand_m3(1.0);
                Constitutive k;
                constitutive_eval(C, F, (real)dt, (real)cs.mu, (real)lam, (real)mass, cs.cls, (real)scale, k);
                m3 gC, gF;
                constitutive_grad(C, F, (real)dt, (real)cs.mu, (real)lam, (real)mass, cs.cls, (real)scale, k, wA, wF, gC, gF);
                for (int which = 0; which < 2; which++) {
                    double maxdiff = 0, maxan = 1e-30;      // norm-wise relative error per 3x3 block
                    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                        m3 Cp = C, Cm = C, Fp = F, Fm = F;
                        double h = which == 0 ? 1e-4 : 1e-7;
                        if (which == 0) { Cp.a[i][j] += h; Cm.a[i][j] -= h; } else { Fp.a[i][j] += h; Fm.a[i][j] -= h; }
                        double fd = (loss_of(Cp, Fp, dt, cs.mu, lam, mass, cs.cls, scale, wA, wF) - loss_of(Cm, Fm, dt, cs.mu, lam, mass, cs.cls, scale, wA, wF)) / (2 * h);
                        double an = which == 0 ? gC.a[i][j] : gF.a[i][j];
                        maxdiff = fmax(maxdiff, fabs(fd - an)); maxan = fmax(maxan, fabs(an));
                    }
                    worst = fmax(worst, maxdiff / maxan);
                }
            }
            printf("%-28s worst rel err %.3g\n", cs.name, worst);
            CHECK(worst < 1e-6, "adjoint mismatch for %s", cs.name);
        }
        // the no-SVD liquid path must agree with the SVD path (same material with a tiny mu)
        for (int t = 0; t < 20; t++) {
            m3 C = rand_m3(5.0), F = m3_add(m3_ident(), rand_m3(0.03));
            Constitutive a, b;
            constitutive_eval(C, F, (real)dt, (real)0.0, (real)lam, (real)mass, FE_MAT_LIQUID_, (real)scale, a);
            constitutive_eval(C, F, (real)dt, (real)1e-300, (real)lam, (real)mass, FE_MAT_LIQUID_, (real)scale, b);
            CHECK(!a.full && b.full, "path selection");
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                CHECK(fabs((double)a.affine.a[i][j] - b.affine.a[i][j]) < 1e-9 * (1e-4 + fabs((double)b.affine.a[i][j])) + 1e-12, "affine differs between SVD / no-SVD path");
                CHECK(fabs((double)a.Fnew.a[i][j] - b.Fnew.a[i][j]) < 1e-7, "Fnew differs between SVD / no-SVD path");
            }
        }
    }
    // ---- boundary + quaternion smoke
    {
        BoundaryP b; b.type = 1; b.lower[0] = 0; b.lower[1] = (real)0.65; b.lower[2] = 0; b.upper[0] = 1; b.upper[1] = (real)0.65; b.upper[2] = 1;
        b.cx = (real)0.5; b.cz = (real)0.5; b.radius = (real)0.42; b.restitution = 0; b.lock_dims = 0;
        real x[3] = {(real)0.99, (real)0.2, (real)0.5}, xn[3], J[3][3];
        boundary_x(b, x, xn, J);
        CHECK(fabs((double)xn[0] - 0.92) < 1e-6 && fabs((double)xn[1] - 0.65) < 1e-7 && J[1][1] == 0, "cylinder impose_x");
        real q[4] = {1, 0, 0, 0}, aa[3] = {0, 0, 0}, qw[4], qo[4], v[3] = {1, 2, 3}, vo[3];
        quat_from_w(aa, qw); quat_mul(qw, q, qo); quat_rotate(v, qo, vo);
        CHECK(fabs((double)vo[0] - 1) < 1e-5 && fabs((double)vo[1] - 2) < 1e-5 && fabs((double)vo[2] - 3) < 1e-5, "identity rotation");
    }
    // ---- J^(1/3), J^(-2/3) of the liquid F update and its adjoint (fe_cbrt_pos: log2 / exp2 + one Newton step in fp32) against pow in fp64
    {
        double worst = 0, worst2 = 0;
        for (int t = 0; t < 20000; t++) {
            const double J = t < 10000 ? 1.0 + 0.2 * (urand() - 0.5) : exp(14.0 * (urand() - 0.5));      // near 1 (any sane state), then six decades
            const real Jr = (real)J;
            const double ulp = sizeof(real) == 4 ? 6e-8 : 1.2e-16;
            worst = fmax(worst, fabs((double)fe_cbrt_pos(Jr) / pow((double)Jr, 1.0 / 3.0) - 1.0) / ulp);
            worst2 = fmax(worst2, fabs((double)fe_pow_m23(Jr) / pow((double)Jr, -2.0 / 3.0) - 1.0) / ulp);
        }
        CHECK(worst <= 2.0 && worst2 <= (sizeof(real) == 4 ? 4.0 : 16.0), "fe_cbrt_pos %g ulp, fe_pow_m23 %g ulp", worst, worst2);      // (fp64: the exponent 1/3 - 1 itself is rounded)
        CHECK(fe_cbrt_pos((real)0) == 0 && std::isnan((double)fe_cbrt_pos((real)-0.5)), "fe_cbrt_pos at 0 / below");
        // the ends of the range: a denormal determinant (the host's log2f does not flush it: the root; the device's does: 0 -- neither is inf or NaN) and an infinite one
        CHECK(std::isfinite((double)fe_cbrt_pos((real)1e-42f)) && (double)fe_cbrt_pos((real)1e-42f) >= 0.0 && (double)fe_cbrt_pos((real)1e-42f) < 1e-10, "fe_cbrt_pos of a denormal: %g", (double)fe_cbrt_pos((real)1e-42f));
        CHECK(std::isinf((double)fe_cbrt_pos((real)INFINITY)) && (double)fe_cbrt_pos((real)INFINITY) > 0, "fe_cbrt_pos of inf: %g", (double)fe_cbrt_pos((real)INFINITY));
    }
    printf("%s (%s): %d failures\n", fails ? "FAILED" : "OK", sizeof(real) == 8 ? "fp64" : "fp32", fails);
    return fails ? 1 : 0;
}
