// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
#include "orc_math.h"
#include <vector>

namespace orc {

void svd3(const M3& F, M3& U, double s[3], M3& V)
{
    M3 G = F;
    for (int i = 0; i < 9; ++i) V.m[i] = 0;
    V(0, 0) = V(1, 1) = V(2, 2) = 1;
    const double eps = 1e-15; // relative orthogonality at which a rotation is rounding noise (1e-16 never settles for ~2 % of inputs)
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += G(i, p) * G(i, p);
                    be += G(i, q) * G(i, q);
                    ga += G(i, p) * G(i, q);
                }
                if (std::fabs(ga) <= eps * std::sqrt(al * be) || ga == 0.0) continue;
                rotated = true;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    double gp = G(i, p), gq = G(i, q);
                    G(i, p) = c * gp - sn * gq;
                    G(i, q) = sn * gp + c * gq;
                    double vp = V(i, p), vq = V(i, q);
                    V(i, p) = c * vp - sn * vq;
                    V(i, q) = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[3];
    for (int j = 0; j < 3; ++j)
        sig[j] = std::sqrt(G(0, j) * G(0, j) + G(1, j) * G(1, j) + G(2, j) * G(2, j));
    int ord[3] = { 0, 1, 2 };
    std::sort(ord, ord + 3, [&](int a, int b) { return sig[a] > sig[b]; });
    M3 Vs, Us;
    for (int j = 0; j < 3; ++j) {
        int o = ord[j];
        s[j] = sig[o];
        for (int i = 0; i < 3; ++i) {
            Vs(i, j) = V(i, o);
            Us(i, j) = (sig[o] > 0) ? G(i, o) / sig[o] : 0.0;
        }
    }
    // rank-deficient completion: rebuild missing columns of U from the others
    double tiny = 1e-300;
    if (s[0] <= tiny) {
        for (int i = 0; i < 9; ++i) Us.m[i] = 0;
        Us(0, 0) = Us(1, 1) = Us(2, 2) = 1;
    }
    else {
        if (s[1] <= tiny * 1e10 || s[1] < 1e-14 * s[0]) {
            // pick any unit vector orthogonal to u0
            double a[3] = { Us(0, 0), Us(1, 0), Us(2, 0) };
            int k = (std::fabs(a[0]) <= std::fabs(a[1]) && std::fabs(a[0]) <= std::fabs(a[2])) ? 0 : (std::fabs(a[1]) <= std::fabs(a[2]) ? 1 : 2);
            double e[3] = { 0, 0, 0 };
            e[k] = 1;
            double d = a[k];
            double u1[3] = { e[0] - d * a[0], e[1] - d * a[1], e[2] - d * a[2] };
            double n = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
            for (int i = 0; i < 3; ++i) Us(i, 1) = u1[i] / n;
        }
        if (s[2] < 1e-14 * s[0]) {
            Us(0, 2) = Us(1, 0) * Us(2, 1) - Us(2, 0) * Us(1, 1);
            Us(1, 2) = Us(2, 0) * Us(0, 1) - Us(0, 0) * Us(2, 1);
            Us(2, 2) = Us(0, 0) * Us(1, 1) - Us(1, 0) * Us(0, 1);
        }
    }
    if (det(Vs) < 0) {
        for (int i = 0; i < 3; ++i) {
            Vs(i, 2) = -Vs(i, 2);
            Us(i, 2) = -Us(i, 2);
        }
    }
    if (det(Us) < 0) {
        for (int i = 0; i < 3; ++i) Us(i, 2) = -Us(i, 2);
        s[2] = -s[2];
    }
    U = Us;
    V = Vs;
}

void sym_eig(int n, const double* Ain, double* w, double* Q)
{
    std::vector<double> A(Ain, Ain + n * n);
    for (int i = 0; i < n * n; ++i) Q[i] = 0;
    for (int i = 0; i < n; ++i) Q[i + n * i] = 1;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                if (i != j) off += A[i + n * j] * A[i + n * j];
                else diag += A[i + n * j] * A[i + n * j];
            }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p + n * q];
                if (apq == 0.0) continue;
                double app = A[p + n * p], aqq = A[q + n * q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { // columns p,q
                    double akp = A[k + n * p], akq = A[k + n * q];
                    A[k + n * p] = c * akp - s * akq;
                    A[k + n * q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) { // rows p,q
                    double apk = A[p + n * k], aqk = A[q + n * k];
                    A[p + n * k] = c * apk - s * aqk;
                    A[q + n * k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double qkp = Q[k + n * p], qkq = Q[k + n * q];
                    Q[k + n * p] = c * qkp - s * qkq;
                    Q[k + n * q] = s * qkp + c * qkq;
                }
            }
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return A[a + n * a] < A[b + n * b]; });
    std::vector<double> Qs(n * n);
    for (int j = 0; j < n; ++j) {
        w[j] = A[ord[j] + n * ord[j]];
        for (int i = 0; i < n; ++i) Qs[i + n * j] = Q[i + n * ord[j]];
    }
    std::memcpy(Q, Qs.data(), sizeof(double) * n * n);
}

void make_pd(int n, double* A)
{
    double w[12], Q[144];
    // symmetrise defensively (callers pass symmetric matrices)
    sym_eig(n, A, w, Q);
    if (w[0] >= 0.0) return;
    for (int i = 0; i < n; ++i) {
        if (w[i] < 0.0) w[i] = 0.0;
        else break;
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += Q[i + n * k] * w[k] * Q[j + n * k];
            A[i + n * j] = s;
        }
}

void make_pd2d(double* m)
{
    const double a = m[0];
    const double b = (m[2] + m[1]) / 2.0;
    const double d = m[3];
    double b2 = b * b;
    const double D = a * d - b2;
    const double T_div_2 = (a + d) / 2.0;
    const double sqrtTT4D = std::sqrt(T_div_2 * T_div_2 - D);
    const double L2 = T_div_2 - sqrtTT4D;
    if (L2 < 0.0) {
        const double L1 = T_div_2 + sqrtTT4D;
        if (L1 <= 0.0) {
            m[0] = m[1] = m[2] = m[3] = 0.0;
        }
        else {
            if (b2 == 0.0) {
                m[0] = L1;
                m[1] = m[2] = m[3] = 0.0;
            }
            else {
                const double L1md = L1 - d;
                const double L1md_div_L1 = L1md / L1;
                m[0] = L1md_div_L1 * L1md;
                m[1] = m[2] = b * L1md_div_L1;
                m[3] = b2 / L1;
            }
        }
    }
}

} // namespace orc
