/*
 * ref_frames_pre.h -- force-included (-include) in front of the REFERENCE's kernel translation unit
 * when oracle/Makefile compiles it for oracle/_ref/libref_frames.so.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's src/raymarcher.cu lines 1..(launch_raymarch - 1), i.e. its #includes and the whole
 * body of raymarch_kernel (:1-174), are piped from /root/reference into g++ unmodified (the last five
 * lines, launch_raymarch with its <<<>>> launch, are nvcc syntax g++ cannot parse).  The CUDA runtime
 * headers are the real ones that ship in this image (triton/backends/nvidia/include).  Nothing of the
 * reference is copied into this repository or left on disk: the text goes through a pipe.
 *
 * What this header adds, and nothing else:
 *   1. threadIdx / blockIdx / blockDim become mutable thread-locals through the hook that
 *      device_launch_parameters.h itself offers (#if !defined(__STORAGE__)); ref_frames.cpp defines
 *      them and sets blockIdx = (x, y), blockDim = (1,1,1), threadIdx = 0 per pixel.
 *   2. tex2D<float4> -- hardware texture filtering, which no CUDA header defines for a host compiler
 *      and the reference's source does not specify -- is declared here and defined in ref_frames.cpp
 *      as the build's documented sampler (oracle: rrto_sky_fetch).  This is the
 *      ONE harness-defined piece of arithmetic in a reference frame.
 *   3. SPIN_A (a literal 0.0f in config.h:21, expanded at geodesics.h:17,41) is re-pointed at a
 *      variable so that a = 0.9 / 0.99 frames come from the same object, as oracle/ref_units.cpp does.
 *   4. Two call-site macros that do not change a single operation: integrate_rk4 is counted (the
 *      per-ray step count the kernel itself does not output), and the two density calls can be
 *      switched to 0.0f for the "skybox only" configuration, which the reference does not have.
 */
#ifndef REF_FRAMES_PRE_H
#define REF_FRAMES_PRE_H

#define __STORAGE__ extern thread_local
#include <cuda_runtime.h>
#include <device_launch_parameters.h>
#include <math.h>

#include "config.h"
extern thread_local float ref_spin_value;
#undef SPIN_A
#define SPIN_A ref_spin_value

template <class T> T tex2D(cudaTextureObject_t tex, float x, float y);
template <> float4 tex2D<float4>(cudaTextureObject_t tex, float x, float y);

#include "raymarcher.h"
#include "math_utils.h"
#include "densities.h"
#include "geodesics.h"
#include "integrators.h"
#include "camera_effects/post_processing.h"

extern thread_local int ref_step_count;
extern thread_local int ref_volumetrics;
#define integrate_rk4(p, v, h) (++ref_step_count, integrate_rk4(p, v, h))
#define getAccretionDensity(p, t) (ref_volumetrics ? getAccretionDensity(p, t) : 0.0f)
#define getDustCloudDensity(p, t) (ref_volumetrics ? getDustCloudDensity(p, t) : 0.0f)

#endif
