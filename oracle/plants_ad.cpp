/*
 * TEST INFRASTRUCTURE ONLY -- the checker's own statement of the acrobot and the two quadrotors.
 *
 * The continuous dynamics (and, for the quaternion plant, the attitude Jacobian and the cost) are written here by
 * hand from the reference's equations, once, as templates over the number type:
 *     control/models/acrobot.py:20-73          -> acrobot_xdot
 *     control/models/Quadrotor_EA.py:47-128    -> quad_ea_xdot
 *     control/models/Quadrotor_Quat.py:61-111  -> quad_quat_xdot
 *     maths/casadi_quaternions.py:5-26         -> quat_normalized / left_matrix / quat_rotation (every function first
 *                                                 normalises q, so the normalisation is differentiated through)
 *     control/models/Quadrotor_Quat.py:142-153 -> quat_cost
 * Derivatives are taken by forward-mode automatic differentiation of that same text (dual numbers; the Hessian of the
 * quaternion cost with duals of duals), which is what CasADi's ca.jacobian does to the reference's graphs.  Nothing
 * here shares text, generator or differentiation method with the plant code of the HIP library (SymPy-generated
 * straight-line code from ilqr_amd/plants.py), so that the two agreeing is evidence about both.
 * Model-level arithmetic stays "parity unpinned" with respect to CasADi itself (not installed; the reference stores
 * no outputs of these plants): the pins are the known answers of tests/test_models.py.
 */
#include <math.h>

namespace {

// ---------------------------------------------------------------- dual numbers: value and N partial derivatives
template <class S, int N>
struct Dual {
    S v;
    S d[N];
};

inline double sin_(double a) { return sin(a); }
inline double cos_(double a) { return cos(a); }
inline double sqrt_(double a) { return sqrt(a); }
inline double lift(double, double c) { return c; }      // the constant c in the number type of the first argument

template <class S, int N> inline Dual<S, N> lift(const Dual<S, N>& like, double c)
{
    Dual<S, N> r;
    r.v = lift(like.v, c);
    for (int i = 0; i < N; ++i) r.d[i] = lift(like.v, 0.0);
    return r;
}
template <class S, int N> inline Dual<S, N> operator+(const Dual<S, N>& a, const Dual<S, N>& b)
{
    Dual<S, N> r;
    r.v = a.v + b.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> operator-(const Dual<S, N>& a, const Dual<S, N>& b)
{
    Dual<S, N> r;
    r.v = a.v - b.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> operator-(const Dual<S, N>& a)
{
    Dual<S, N> r;
    r.v = -a.v;
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> operator*(const Dual<S, N>& a, const Dual<S, N>& b)
{
    Dual<S, N> r;
    r.v = a.v * b.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> operator/(const Dual<S, N>& a, const Dual<S, N>& b)
{
    Dual<S, N> r;
    r.v = a.v / b.v;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
template <class S, int N> inline Dual<S, N> operator*(double c, const Dual<S, N>& a)
{
    Dual<S, N> r;
    r.v = c * a.v;
    for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> operator*(const Dual<S, N>& a, double c) { return c * a; }
template <class S, int N> inline Dual<S, N> operator/(const Dual<S, N>& a, double c)
{
    Dual<S, N> r;
    r.v = a.v / c;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / c;
    return r;
}
template <class S, int N> inline Dual<S, N> operator+(const Dual<S, N>& a, double c)
{
    Dual<S, N> r = a;
    r.v = a.v + c;
    return r;
}
template <class S, int N> inline Dual<S, N> operator+(double c, const Dual<S, N>& a) { return a + c; }
template <class S, int N> inline Dual<S, N> operator-(const Dual<S, N>& a, double c)
{
    Dual<S, N> r = a;
    r.v = a.v - c;
    return r;
}
template <class S, int N> inline Dual<S, N> operator-(double c, const Dual<S, N>& a) { return (-a) + c; }
template <class S, int N> inline Dual<S, N> sin_(const Dual<S, N>& a)
{
    Dual<S, N> r;
    const S c = cos_(a.v);
    r.v = sin_(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i];
    return r;
}
template <class S, int N> inline Dual<S, N> cos_(const Dual<S, N>& a)
{
    Dual<S, N> r;
    const S s = sin_(a.v);
    r.v = cos_(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = -(s * a.d[i]);
    return r;
}
template <class S, int N> inline Dual<S, N> sqrt_(const Dual<S, N>& a)
{
    Dual<S, N> r;
    r.v = sqrt_(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / (2.0 * r.v);
    return r;
}

// ---------------------------------------------------------------- acrobot (acrobot.py:20-73)
template <class T>
void acrobot_xdot(const T* x, const T* u, T* f)
{
    const double m1 = 1, m2 = 1, l1 = 1, lc1 = 0.5, lc2 = 1, Ic1 = 0.083, Ic2 = 0.33, b1 = 0.1, b2 = 0.1, g = 9.81;
    const T th1 = x[0], th2 = x[1], th1_dot = x[2], th2_dot = x[3];
    const T s1 = sin_(th1), s2 = sin_(th2), s12 = sin_(th1 + th2), c2 = cos_(th2);
    const double m2l1lc2 = m2 * l1 * lc2;
    const double I1 = Ic1 + m1 * lc1 * lc1, I2 = Ic2 + m2 * lc2 * lc2;
    const T M11 = (I1 + I2 + m2 * l1 * l1) + (2 * m2l1lc2) * c2;
    const T M12 = I2 + m2l1lc2 * c2;
    const T M22 = lift(th1, I2);
    const T Coreolis1 = (-2 * m2l1lc2) * s2 * th2_dot * th1_dot - m2l1lc2 * s2 * (th2_dot * th2_dot);
    const T Coreolis2 = m2l1lc2 * s2 * (th1_dot * th1_dot);
    const T Grav1 = (g * m1 * lc1) * s1 + (g * m2) * (l1 * s1 + lc2 * s12);
    const T Grav2 = (g * m2 * lc2) * s12;
    const T Bias1 = Coreolis1 + Grav1 + b1 * th1_dot;
    const T Bias2 = Coreolis2 + Grav2 + b2 * th2_dot;
    // solve(M, B u - Bias), B = [0; 1]
    const T r1 = -Bias1, r2 = u[0] - Bias2;
    const T det = M11 * M22 - M12 * M12;
    f[0] = th1_dot;
    f[1] = th2_dot;
    f[2] = (M22 * r1 - M12 * r2) / det;
    f[3] = (M11 * r2 - M12 * r1) / det;
}

// ---------------------------------------------------------------- thrust, moments, rigid body (both quadrotors)
constexpr double QL = 0.15, QKF = 1.0, QKM = 0.0245, QM = 0.775, QIX = 0.0015, QIY = 0.0025, QIZ = 0.0035;

// alpha = R I^-1 (tau - w x (I w)), thrust along body z
template <class T>
void body_wrench(const T* u, const T* w, T* ia /* I^-1 (tau - w x I w) */, T& thrust)
{
    const T f0 = QKF * u[0], f1 = QKF * u[1], f2 = QKF * u[2], f3 = QKF * u[3];
    thrust = f0 + f1 + f2 + f3;
    const T Mx = QL * (f1 - f3), My = QL * (f2 - f0);
    const T Mz = QKM * u[0] - QKM * u[1] + QKM * u[2] - QKM * u[3];
    const T Iw0 = QIX * w[0], Iw1 = QIY * w[1], Iw2 = QIZ * w[2];
    const T c0 = w[1] * Iw2 - w[2] * Iw1, c1 = w[2] * Iw0 - w[0] * Iw2, c2 = w[0] * Iw1 - w[1] * Iw0;
    ia[0] = (Mx - c0) / QIX;
    ia[1] = (My - c1) / QIY;
    ia[2] = (Mz - c2) / QIZ;
}

// ---------------------------------------------------------------- Euler-angle quadrotor (Quadrotor_EA.py:47-128)
// state [p(3), yaw, pitch, roll, v(3), yaw', pitch', roll']
template <class T>
void quad_ea_xdot(const T* x, const T* u, T* f)
{
    const T y = x[3], p = x[4], r = x[5];
    const T rpy_rate[3] = {x[11], x[10], x[9]};
    const T cr = cos_(r), cp = cos_(p), cy = cos_(y), sr = sin_(r), sp = sin_(p), sy = sin_(y);
    const T R[3][3] = {{cp * cy, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},
                       {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                       {-sp, cp * sr, cr * cp}};
    // w_B = M [roll', pitch', yaw'], M = [[1, 0, -sp], [0, cr, sr cp], [0, -sr, cr cp]]
    const T w[3] = {rpy_rate[0] - sp * rpy_rate[2], cr * rpy_rate[1] + sr * cp * rpy_rate[2],
                    cr * cp * rpy_rate[2] - sr * rpy_rate[1]};
    T ia[3], thrust;
    body_wrench(u, w, ia, thrust);
    T alpha[3];
    for (int i = 0; i < 3; ++i) alpha[i] = R[i][0] * ia[0] + R[i][1] * ia[1] + R[i][2] * ia[2];
    const T yaw_rate = x[9], pitch_rate = x[10];
    // Mdt [roll', pitch', yaw']
    const T md[3] = {(-(cy * sp) * pitch_rate - sy * cp * yaw_rate) * rpy_rate[0] - cy * yaw_rate * rpy_rate[1],
                     (cy * cp * yaw_rate - sy * sp * pitch_rate) * rpy_rate[0] - sy * yaw_rate * rpy_rate[1],
                     -(cp * pitch_rate) * rpy_rate[0]};
    const T q[3] = {alpha[0] - md[0], alpha[1] - md[1], alpha[2] - md[2]};
    // Minv = [[cy/cp, sy/cp, 0], [-sy, cy, 0], [cy/cp sp, sy/cp sp, 1]]
    const T acc_r = (cy / cp) * q[0] + (sy / cp) * q[1];
    const T acc_p = cy * q[1] - sy * q[0];
    const T acc_y = (cy / cp) * sp * q[0] + (sy / cp) * sp * q[1] + q[2];
    f[0] = x[6]; f[1] = x[7]; f[2] = x[8];
    f[3] = x[9]; f[4] = x[10]; f[5] = x[11];
    f[6] = (R[0][2] * thrust) / QM;
    f[7] = (R[1][2] * thrust) / QM;
    f[8] = (R[2][2] * thrust - QM * 9.81) / QM;
    f[9] = acc_y; f[10] = acc_p; f[11] = acc_r;
}

// ---------------------------------------------------------------- quaternion algebra (casadi_quaternions.py:5-26)
template <class T>
void quat_normalized(const T* q, T* n)
{
    const T norm = sqrt_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) n[i] = q[i] / norm;
}
// G(q) = L(q^) H: the last three columns of the left matrix, 4 x 3
template <class T>
void quat_G(const T* q, T G[4][3])
{
    T n[4];
    quat_normalized(q, n);
    const T w = n[0], x = n[1], y = n[2], z = n[3];
    G[0][0] = -x; G[0][1] = -y; G[0][2] = -z;
    G[1][0] = w;  G[1][1] = -z; G[1][2] = y;
    G[2][0] = z;  G[2][1] = w;  G[2][2] = -x;
    G[3][0] = -y; G[3][1] = x;  G[3][2] = w;
}
// R = H' T L T L H
template <class T>
void quat_rotation(const T* q, T R[3][3])
{
    T n[4];
    quat_normalized(q, n);
    const T w = n[0], x = n[1], y = n[2], z = n[3];
    const T L[4][4] = {{w, -x, -y, -z}, {x, w, -z, y}, {y, z, w, -x}, {z, -y, x, w}};
    const double t[4] = {1.0, -1.0, -1.0, -1.0};
    // (T L T L)[i][j] = sum_k t_i L[i][k] t_k L[k][j]; rows and columns 1..3
    for (int i = 1; i < 4; ++i)
        for (int j = 1; j < 4; ++j) {
            T s = (t[i] * t[0]) * (L[i][0] * L[0][j]);
            for (int k = 1; k < 4; ++k) s = s + (t[i] * t[k]) * (L[i][k] * L[k][j]);
            R[i - 1][j - 1] = s;
        }
}

// ---------------------------------------------------------------- quaternion quadrotor (Quadrotor_Quat.py:61-111)
// state [q(4), p(3), omega(3) world, v(3)]
template <class T>
void quad_quat_xdot(const T* x, const T* u, T* f)
{
    T R[3][3], G[4][3];
    quat_rotation(x, R);
    quat_G(x, G);
    T w[3];
    for (int i = 0; i < 3; ++i) w[i] = R[0][i] * x[7] + R[1][i] * x[8] + R[2][i] * x[9];      // R' omega
    for (int i = 0; i < 4; ++i) f[i] = 0.5 * (G[i][0] * w[0] + G[i][1] * w[1] + G[i][2] * w[2]);
    T ia[3], thrust;
    body_wrench(u, w, ia, thrust);
    f[4] = x[10]; f[5] = x[11]; f[6] = x[12];
    for (int i = 0; i < 3; ++i) f[7 + i] = R[i][0] * ia[0] + R[i][1] * ia[1] + R[i][2] * ia[2];
    f[10] = (R[0][2] * thrust) / QM;
    f[11] = (R[1][2] * thrust) / QM;
    f[12] = (R[2][2] * thrust - QM * 9.81) / QM;
}

// ((0.5 e)' W) e for a D x D row-major matrix of plain numbers
template <class T, int D>
T half_form(const double* W, const T* e)
{
    T total = lift(e[0], 0.0);
    for (int j = 0; j < D; ++j) {
        T t = W[0 * D + j] * (0.5 * e[0]);
        for (int i = 1; i < D; ++i) t = t + W[i * D + j] * (0.5 * e[i]);
        total = total + t * e[j];
    }
    return total;
}
// 0.5 e' Q e + 0.5 du' R du, e = Gx(x)' (x - xref), Q [12 x 12] and R [4 x 4] any matrices
// (Quadrotor_Quat.py:142-153)
template <class T>
T quat_cost(const T* x, const T* u, const double* xref, const double* uref, const double* Q, const double* R)
{
    T G[4][3];
    quat_G(x, G);
    T e[12], du[4];
    for (int j = 0; j < 3; ++j) {
        e[j] = G[0][j] * (x[0] - xref[0]);
        for (int i = 1; i < 4; ++i) e[j] = e[j] + G[i][j] * (x[i] - xref[i]);
    }
    for (int j = 0; j < 9; ++j) e[3 + j] = x[4 + j] - xref[4 + j];
    for (int j = 0; j < 4; ++j) du[j] = u[j] - uref[j];
    return half_form<T, 12>(Q, e) + half_form<T, 4>(R, du);
}

// ---------------------------------------------------------------- Jacobians by forward mode
template <int NX, int NU, void (*F)(const Dual<double, NX + NU>*, const Dual<double, NX + NU>*, Dual<double, NX + NU>*)>
void jacobian(const double* x, const double* u, double* f, double* A, double* B)
{
    typedef Dual<double, NX + NU> D;
    D xs[NX], us[NU], fs[NX];
    for (int i = 0; i < NX; ++i) {
        xs[i].v = x[i];
        for (int k = 0; k < NX + NU; ++k) xs[i].d[k] = (k == i) ? 1.0 : 0.0;
    }
    for (int i = 0; i < NU; ++i) {
        us[i].v = u[i];
        for (int k = 0; k < NX + NU; ++k) us[i].d[k] = (k == NX + i) ? 1.0 : 0.0;
    }
    F(xs, us, fs);
    for (int i = 0; i < NX; ++i) {
        f[i] = fs[i].v;
        for (int j = 0; j < NX; ++j) A[i * NX + j] = fs[i].d[j];
        for (int j = 0; j < NU; ++j) B[i * NU + j] = fs[i].d[NX + j];
    }
}

}  // namespace

extern "C" {

void oracle_acrobot_xdot(const double* x, const double* u, double* f) { acrobot_xdot<double>(x, u, f); }
void oracle_acrobot_xdot_jac(const double* x, const double* u, double* f, double* A, double* B)
{
    jacobian<4, 1, acrobot_xdot<Dual<double, 5> > >(x, u, f, A, B);
}
void oracle_quad_ea_xdot(const double* x, const double* u, double* f) { quad_ea_xdot<double>(x, u, f); }
void oracle_quad_ea_xdot_jac(const double* x, const double* u, double* f, double* A, double* B)
{
    jacobian<12, 4, quad_ea_xdot<Dual<double, 16> > >(x, u, f, A, B);
}
void oracle_quad_quat_xdot(const double* x, const double* u, double* f) { quad_quat_xdot<double>(x, u, f); }
void oracle_quad_quat_xdot_jac(const double* x, const double* u, double* f, double* A, double* B)
{
    jacobian<13, 4, quad_quat_xdot<Dual<double, 17> > >(x, u, f, A, B);
}
/* G(q) [4 x 3], row-major */
void oracle_quad_quat_G(const double* x, double* G)
{
    double g[4][3];
    quat_G<double>(x, g);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) G[i * 3 + j] = g[i][j];
}
/* w = [xref(13), uref(4), Q(12 x 12), R(4 x 4)] */
void oracle_quad_quat_cost(const double* w, const double* x, const double* u, double* c)
{
    *c = quat_cost<double>(x, u, w, w + 13, w + 17, w + 161);
}
/* l_x [13], l_u [4], l_xx [13 x 13] (l_ux = 0, l_uu = (R + R')/2) */
void oracle_quad_quat_cost_derivs(const double* w, const double* x, const double* u, double* lx, double* lu, double* lxx)
{
    typedef Dual<double, 13> D1;
    typedef Dual<D1, 13> D2;
    D2 xs[13], us[4];
    for (int i = 0; i < 13; ++i) {
        xs[i].v.v = x[i];
        for (int k = 0; k < 13; ++k) {
            xs[i].v.d[k] = (k == i) ? 1.0 : 0.0;
            xs[i].d[k].v = (k == i) ? 1.0 : 0.0;
            for (int t = 0; t < 13; ++t) xs[i].d[k].d[t] = 0.0;
        }
    }
    for (int i = 0; i < 4; ++i) {
        us[i].v.v = u[i];
        for (int k = 0; k < 13; ++k) {
            us[i].v.d[k] = 0.0;
            us[i].d[k].v = 0.0;
            for (int t = 0; t < 13; ++t) us[i].d[k].d[t] = 0.0;
        }
    }
    const D2 c = quat_cost<D2>(xs, us, w, w + 13, w + 17, w + 161);
    for (int i = 0; i < 13; ++i) {
        lx[i] = c.d[i].v;
        for (int j = 0; j < 13; ++j) lxx[i * 13 + j] = c.d[i].d[j];
    }
    const double* Rm = w + 161;
    for (int j = 0; j < 4; ++j) {
        lu[j] = 0.0;
        for (int k = 0; k < 4; ++k) lu[j] += 0.5 * (Rm[j * 4 + k] + Rm[k * 4 + j]) * (u[k] - w[13 + k]);
    }
}

}  /* extern "C" */
