// sphere_bin.hpp — the scalar bin formula of build_spherical_kernel (tf_ops/buildkernel/tf_buildkernel_gpu.cu:49-74),
// shared by buildkernel.hip and the fused graph construction of nnquery.hip.  Must match oracle_sphere_bin bit for bit.
#pragma once
#include "../../include/sph3d_atan2f.h"

namespace sph3d {

#define SPH3D_PI 3.14159265358979323846   // double, == glibc M_PI

template <bool OCML>
__device__ __forceinline__ int sphere_bin(float dx, float dy, float dz, float dist, float radius, int n, int p, int q)
{
    const float M_EPSf = 1.01e-3F;                                     // tf_buildkernel_gpu.cu:5-7
    float dist2D = dx * dx + dy * dy;                                  // :49
    dist2D = sqrtf(dist2D);                                            // :50
    if (!(dist > M_EPSf && (double)fabsf(dist - M_EPSf) > 1e-6)) return 0;   // :52-53 self / coincident
    float theta = OCML ? atan2f(dy, dx) : sph3d_atan2f(dy, dx);        // :55
    float phi = OCML ? atan2f(dz, dist2D) : sph3d_atan2f(dz, dist2D);  // :56
    theta = (float)((double)theta < SPH3D_PI ? (double)theta : -SPH3D_PI);        // :58
    theta = (float)((double)theta > -SPH3D_PI ? (double)theta : -SPH3D_PI);       // :59
    theta = (float)((double)theta + SPH3D_PI);                                    // :60
    phi = (float)((double)phi < (SPH3D_PI / 2) ? (double)phi : (SPH3D_PI / 2));   // :62
    phi = (float)((double)phi > (-SPH3D_PI / 2) ? (double)phi : (-SPH3D_PI / 2)); // :63
    phi = (float)((double)phi + SPH3D_PI / 2);                                    // :64
    const float alpha = (float)((double)((theta * (float)n) / 2.0f) / SPH3D_PI);  // :66
    const float beta = (float)((double)(phi * (float)p) / SPH3D_PI);              // :67
    const float gamma = (dist * (float)q) / (radius + 1e-6F);                     // :68
    int nID = (int)alpha; nID = nID < n - 1 ? nID : n - 1;                        // :70-72
    int pID = (int)beta;  pID = pID < p - 1 ? pID : p - 1;
    int qID = (int)gamma; qID = qID < q - 1 ? qID : q - 1;
    return qID * p * n + pID * n + nID + 1;                                       // :74
}

}  // namespace sph3d
