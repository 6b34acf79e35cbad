// Device-side rotation math shared by the pose / kinematics / projection kernels.
// Formulas follow the reference operators line by line so that the branch structure and the
// epsilon handling are identical (parity at 1e-4 relative needs that near theta ~ 0 and ~ pi):
//   rot6d_to_rotmat                 hmr/geometry.py:47-61
//   rotation_matrix_to_quaternion   hmr/geometry.py:266-346
//   quaternion_to_angle_axis        hmr/geometry.py:213-263
//   batch_rodrigues / quat_to_rotmat hmr/geometry.py:9-45
// Backward functions are hand-derived adjoints of exactly those expression graphs.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define NEMO_NORM_EPS 1e-12f   // F.normalize eps

// x[6] -> R[9] (row-major).  x is read as a 3x2 matrix: a1 = (x0,x2,x4), a2 = (x1,x3,x5).
__device__ __forceinline__ void rot6d_fwd(const float* x, float* R) {
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), NEMO_NORM_EPS);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), NEMO_NORM_EPS);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2],
                         b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
    for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = b1[r]; R[r * 3 + 1] = b2[r]; R[r * 3 + 2] = b3[r]; }
}

// dR[9] -> dx[6] (overwrites dx).
__device__ __forceinline__ void rot6d_bwd(const float* x, const float* dR, float* dx) {
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float r1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    const float n1 = fmaxf(r1, NEMO_NORM_EPS);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float r2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float n2 = fmaxf(r2, NEMO_NORM_EPS);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float db1[3] = {dR[0], dR[3], dR[6]}, db2[3] = {dR[1], dR[4], dR[7]};
    const float db3[3] = {dR[2], dR[5], dR[8]};
    // b3 = b1 x b2:  db1 += b2 x db3,  db2 += db3 x b1
    db1[0] += b2[1] * db3[2] - b2[2] * db3[1];
    db1[1] += b2[2] * db3[0] - b2[0] * db3[2];
    db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
    db2[0] += db3[1] * b1[2] - db3[2] * b1[1];
    db2[1] += db3[2] * b1[0] - db3[0] * b1[2];
    db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
    // b2 = u / max(|u|, eps)
    float du[3];
    if (r2 > NEMO_NORM_EPS) {
        const float s = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) du[i] = (db2[i] - b2[i] * s) / n2;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) du[i] = db2[i] / n2;
    }
    // u = a2 - d b1,  d = b1 . a2
    const float dd = -(du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2]);
    float da2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        da2[i] = du[i] + dd * b1[i];
        db1[i] += -d * du[i] + dd * a2[i];
    }
    float da1[3];
    if (r1 > NEMO_NORM_EPS) {
        const float s = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) da1[i] = (db1[i] - b1[i] * s) / n1;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) da1[i] = db1[i] / n1;
    }
    dx[0] = da1[0]; dx[2] = da1[1]; dx[4] = da1[2];
    dx[1] = da2[0]; dx[3] = da2[1]; dx[5] = da2[2];
}

// Selected quaternion numerator qs[4] (w,x,y,z order), its normaliser t, and the branch id.
__device__ __forceinline__ int rotmat_quat_branch(const float* R, float* qs, float* t) {
    // reference works on m = R^T: m[i][j] = R[j][i]
    const float m00 = R[0], m11 = R[4], m22 = R[8];
    const float m01 = R[3], m10 = R[1], m02 = R[6], m20 = R[2], m12 = R[7], m21 = R[5];
    const bool d2 = m22 < 1e-6f, d01 = m00 > m11, d0n1 = m00 < -m11;
    if (d2 && d01) {
        *t = 1.f + m00 - m11 - m22;
        qs[0] = m12 - m21; qs[1] = *t; qs[2] = m01 + m10; qs[3] = m20 + m02; return 0;
    } else if (d2) {
        *t = 1.f - m00 + m11 - m22;
        qs[0] = m20 - m02; qs[1] = m01 + m10; qs[2] = *t; qs[3] = m12 + m21; return 1;
    } else if (d0n1) {
        *t = 1.f - m00 - m11 + m22;
        qs[0] = m01 - m10; qs[1] = m20 + m02; qs[2] = m12 + m21; qs[3] = *t; return 2;
    }
    *t = 1.f + m00 + m11 + m22;
    qs[0] = *t; qs[1] = m12 - m21; qs[2] = m20 - m02; qs[3] = m01 - m10; return 3;
}

// R[9] -> aa[3].
__device__ __forceinline__ void rotmat_to_aa_fwd(const float* R, int zero_nan, float* aa) {
    float qs[4], t;
    rotmat_quat_branch(R, qs, &t);
    const float st = sqrtf(t);
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (qs[i] / st) * 0.5f;
    const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const float s = sqrtf(s2), c = q[0];
    const float tt = 2.0f * (c < 0.f ? atan2f(-s, -c) : atan2f(s, c));
    const float k = s2 > 0.f ? tt / s : 2.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = q[i + 1] * k;
        if (zero_nan && isnan(v)) v = 0.f;
        aa[i] = v;
    }
}

// g[3] = d/d aa  ->  dR[9] += adjoint.   (accumulates into dR)
__device__ __forceinline__ void rotmat_to_aa_bwd(const float* R, int zero_nan, const float* g_in, float* dR) {
    float qs[4], t;
    const int br = rotmat_quat_branch(R, qs, &t);
    const float st = sqrtf(t);
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (qs[i] / st) * 0.5f;
    const float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const float s = sqrtf(s2), c = q[0];
    const float tt = 2.0f * (c < 0.f ? atan2f(-s, -c) : atan2f(s, c));
    const float k = s2 > 0.f ? tt / s : 2.0f;
    float g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g[i] = g_in[i];
        if (zero_nan && isnan(q[i + 1] * k)) g[i] = 0.f;   // index_put_(isnan, 0) backward
    }
    // aa_i = q_i * k(s, c)
    const float gk = g[0] * q[1] + g[1] * q[2] + g[2] * q[3];
    float dq[4];
    if (s2 > 0.f) {
        const float r2 = s2 + c * c;
        const float dk_ds = (2.0f * c / r2) / s - tt / s2;
        const float dk_dc = (-2.0f * s / r2) / s;
        const float ds = gk * dk_ds;
        dq[0] = gk * dk_dc;
#pragma unroll
        for (int i = 0; i < 3; ++i) dq[i + 1] = g[i] * k + ds * (q[i + 1] / s);
    } else {
        // autograd: grad of the unselected k_pos = 0/0 branch is NaN (matches the reference, which
        // is why its author could not initialise the last layer to exactly zero, :121-123)
        const float nanv = __builtin_nanf("");
        dq[0] = gk * nanv;
#pragma unroll
        for (int i = 0; i < 3; ++i) dq[i + 1] = g[i] * k + gk * nanv;
    }
    // q = 0.5 * qs / sqrt(t)
    float dqs[4];
    float dt = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dqs[i] = 0.5f * dq[i] / st;
        dt += dq[i] * qs[i];
    }
    dt = -0.25f * dt / (t * st);
    // scatter to R (m = R^T: m01 = R[3], m10 = R[1], m02 = R[6], m20 = R[2], m12 = R[7], m21 = R[5])
    if (br == 0) {
        dt += dqs[1];
        dR[0] += dt; dR[4] -= dt; dR[8] -= dt;
        dR[7] += dqs[0]; dR[5] -= dqs[0];
        dR[3] += dqs[2]; dR[1] += dqs[2];
        dR[2] += dqs[3]; dR[6] += dqs[3];
    } else if (br == 1) {
        dt += dqs[2];
        dR[0] -= dt; dR[4] += dt; dR[8] -= dt;
        dR[2] += dqs[0]; dR[6] -= dqs[0];
        dR[3] += dqs[1]; dR[1] += dqs[1];
        dR[7] += dqs[3]; dR[5] += dqs[3];
    } else if (br == 2) {
        dt += dqs[3];
        dR[0] -= dt; dR[4] -= dt; dR[8] += dt;
        dR[3] += dqs[0]; dR[1] -= dqs[0];
        dR[2] += dqs[1]; dR[6] += dqs[1];
        dR[7] += dqs[2]; dR[5] += dqs[2];
    } else {
        dt += dqs[0];
        dR[0] += dt; dR[4] += dt; dR[8] += dt;
        dR[7] += dqs[1]; dR[5] -= dqs[1];
        dR[2] += dqs[2]; dR[6] -= dqs[2];
        dR[3] += dqs[3]; dR[1] -= dqs[3];
    }
}

// unit quaternion (w,x,y,z) -> R[9]
__device__ __forceinline__ void quat_to_R(float w, float x, float y, float z, float* R) {
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;    R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;    R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
}

// theta[3] -> R[9], quaternion form (hmr/geometry.py:9-45)
__device__ __forceinline__ void rodrigues_fwd(const float* th, float* R) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float angle = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float ax = th[0] / angle, ay = th[1] / angle, az = th[2] / angle;
    const float half = angle * 0.5f;
    const float cw = cosf(half), sw = sinf(half);
    float q[4] = {cw, sw * ax, sw * ay, sw * az};
    const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    quat_to_R(q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq, R);
}

// G[9] = dL/dR -> dth[3] (overwrites)
__device__ __forceinline__ void rodrigues_bwd(const float* th, const float* G, float* dth) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float angle = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float ax[3] = {th[0] / angle, th[1] / angle, th[2] / angle};
    const float half = angle * 0.5f;
    const float cw = cosf(half), sw = sinf(half);
    const float qr[4] = {cw, sw * ax[0], sw * ax[1], sw * ax[2]};
    const float nq = sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
    const float w = qr[0] / nq, x = qr[1] / nq, y = qr[2] / nq, z = qr[3] / nq;
    float dqn[4];
    dqn[0] = 2 * w * (G[0] + G[4] + G[8]) + 2 * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    dqn[1] = 2 * x * (G[0] - G[4] - G[8]) + 2 * (y * G[1] + z * G[2] + y * G[3] - w * G[5] + z * G[6] + w * G[7]);
    dqn[2] = 2 * y * (-G[0] + G[4] - G[8]) + 2 * (x * G[1] + w * G[2] + x * G[3] + z * G[5] - w * G[6] + z * G[7]);
    dqn[3] = 2 * z * (-G[0] - G[4] + G[8]) + 2 * (-w * G[1] + x * G[2] + w * G[3] + y * G[5] + x * G[6] + y * G[7]);
    const float qn[4] = {w, x, y, z};
    const float dot = qn[0] * dqn[0] + qn[1] * dqn[1] + qn[2] * dqn[2] + qn[3] * dqn[3];
    float dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = (dqn[i] - qn[i] * dot) / nq;
    // q = [cos(half), sin(half) * axis]
    float dhalf = -sw * dq[0] + cw * (ax[0] * dq[1] + ax[1] * dq[2] + ax[2] * dq[3]);
    const float dax[3] = {sw * dq[1], sw * dq[2], sw * dq[3]};
    float dangle = 0.5f * dhalf;
    // axis = theta / angle
    dangle += -(dax[0] * th[0] + dax[1] * th[1] + dax[2] * th[2]) / (angle * angle);
    // angle = || theta + 1e-8 ||
    dth[0] = dax[0] / angle + dangle * t0 / angle;
    dth[1] = dax[1] / angle + dangle * t1 / angle;
    dth[2] = dax[2] / angle + dangle * t2 / angle;
}

// matrix-form Rodrigues of the vendored lbs (human_body_prior/body_model/lbs.py:303-334), fwd only
__device__ __forceinline__ void rodrigues_lbs_fwd(const float* th, float* R) {
    const float t0 = th[0] + 1e-8f, t1 = th[1] + 1e-8f, t2 = th[2] + 1e-8f;
    const float angle = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    const float rx = th[0] / angle, ry = th[1] / angle, rz = th[2] / angle;
    const float s = sinf(angle), c1 = 1.f - cosf(angle);
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float kk = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) kk += K[r * 3 + m] * K[m * 3 + cc];
            R[r * 3 + cc] = (r == cc ? 1.f : 0.f) + s * K[r * 3 + cc] + c1 * kk;
        }
}
