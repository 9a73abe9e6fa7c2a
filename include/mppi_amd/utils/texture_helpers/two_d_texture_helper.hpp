/**
 * two_d_texture_helper.hpp — generic 2-D map lookup for Dynamics / Cost plugins (elevation maps, cost maps): world pose ->
 * map frame -> normalised texture coordinate -> bilinear (or nearest) sample with clamp or border addressing.
 *
 * Replaces the reference's TextureHelper / TwoDTextureHelper (include/mppi/utils/texture_helpers/texture_helper.cuh:17-204,
 * texture_helper.cu:94-133, 270-289; two_d_texture_helper.cu:66-73 and 151-245).  The reference samples with the CUDA
 * texture unit (tex2D, cudaFilterModeLinear: 9-bit fixed-point interpolation weights) on the device and with an fp32
 * restatement (queryTextureCPU) on the host, and its own tests accept the difference.  gfx950 compute kernels have no
 * reason to route a few KB of L2-resident map through the image path: this helper evaluates the reference's HOST
 * formula in fp32 on the device too, so device, host and oracle agree bit for bit and the interpolation is exact at the
 * cell centres.
 *
 * Layout: texel (row r, column c) of texture i at data[(r * width + c) * NC + ch], NC channels per texel (1 or 4 in the
 * reference: float / float4).  Same method names and argument meaning as the reference; float3 arguments are plain
 * float[3].  The object is a POD that travels as a kernel argument like every other plugin member.
 */
#ifndef MPPI_AMD_TWO_D_TEXTURE_HELPER_HPP_
#define MPPI_AMD_TWO_D_TEXTURE_HELPER_HPP_

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPPI_TEX_HD __host__ __device__
#else
#define MPPI_TEX_HD
#endif

namespace mppi
{
namespace texture
{
enum AddressMode : int
{
  ADDRESS_CLAMP = 0,   ///< cudaAddressModeClamp (the reference's default, texture_helper.cuh:41-43)
  ADDRESS_BORDER = 1,  ///< cudaAddressModeBorder: border_color outside the map
};
enum FilterMode : int
{
  FILTER_LINEAR = 0,  ///< cudaFilterModeLinear (default, texture_helper.cuh:48)
  FILTER_POINT = 1,
};

/** reference: TextureParams (texture_helper.cuh:17-63) without the CUDA array / texture object */
struct TextureParams2D
{
  const float* data = nullptr;  ///< [height][width][NC], device (or host, for the host-side call) pointer
  int width = 0, height = 0;
  int use = 0;                  ///< checkTextureUse()
  int address_mode[2] = { ADDRESS_CLAMP, ADDRESS_CLAMP };
  int filter_mode = FILTER_LINEAR;
  float border_color[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  float origin[3] = { 0.0f, 0.0f, 0.0f };
  float rotations[9] = { 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f };  ///< row-major 3x3
  float resolution[3] = { 1.0f, 1.0f, 1.0f };                                     ///< metres per texel
};

template <int NUM_TEXTURES, int NC = 1>
struct TwoDTextureHelper
{
  static constexpr int CHANNELS = NC;
  TextureParams2D textures_[NUM_TEXTURES];

  MPPI_TEX_HD inline bool checkTextureUse(const int index) const
  {
    return textures_[index].use != 0;
  }

  /** texture_helper.cu:94-104 */
  MPPI_TEX_HD inline void worldPoseToMapPose(const int index, const float* input, float* output) const
  {
    const TextureParams2D& p = textures_[index];
    const float dx = input[0] - p.origin[0], dy = input[1] - p.origin[1], dz = input[2] - p.origin[2];
    const float* R = p.rotations;
    output[0] = R[0] * dx + R[1] * dy + R[2] * dz;
    output[1] = R[3] * dx + R[4] * dy + R[5] * dz;
    output[2] = R[6] * dx + R[7] * dy + R[8] * dz;
  }

  /** texture_helper.cu:106-124: metres -> texels -> normalised coordinate (the 2-D helper has depth 0: z only scaled) */
  MPPI_TEX_HD inline void mapPoseToTexCoord(const int index, const float* input, float* output) const
  {
    const TextureParams2D& p = textures_[index];
    output[0] = (input[0] / p.resolution[0]) / (float)p.width;
    output[1] = (input[1] / p.resolution[1]) / (float)p.height;
    output[2] = input[2] / p.resolution[2];
  }

  /**
   * The four taps and weights of one bilinear lookup (two_d_texture_helper.cu:151-245, queryTextureCPU): normalised
   * coordinate -> texel units, minus half a cell (values sit at the cell centres), addressing, the 2 x 2 neighbourhood.
   * Branch-free: in border mode an outside point is only FLAGGED and its taps are those of the clamped coordinate (valid
   * addresses; the caller substitutes the border colour), so that a caller can compute several footprints, issue all their
   * loads, and only then combine — see queryTextureAtWorldPoseBatch().
   */
  struct LinearFootprint
  {
    int i11, i12, i21, i22;  ///< texel indices (row-major, before the channel stride): (y_min, x_min), (y_min, x_max), (y_max, ..)
    float wx0, wx1, wy0, wy1;
    bool border;
  };
  MPPI_TEX_HD inline LinearFootprint linearFootprint(const int index, const float* point) const
  {
    const TextureParams2D& p = textures_[index];
    const int w = p.width, h = p.height;
    float qx = point[0] * (float)w - 0.5f;
    float qy = point[1] * (float)h - 0.5f;
    // a NaN coordinate (a diverged rollout) would reach (int)floorf(NaN); it samples texel 0 like ARStandardCost's lookup
    qx = (qx == qx) ? qx : 0.0f;
    qy = (qy == qy) ? qy : 0.0f;
    LinearFootprint f;
    const bool out_x = (qx > (float)(w - 1)) || (qx <= 0.0f), out_y = (qy > (float)(h - 1)) || (qy <= 0.0f);
    f.border = (p.address_mode[0] != ADDRESS_CLAMP && out_x) || (p.address_mode[1] != ADDRESS_CLAMP && out_y);
    // clamp addressing; for an inside point of border addressing these are the identity
    qx = (qx > (float)(w - 1)) ? (float)(w - 1) : ((qx <= 0.0f) ? 0.0f : qx);
    qy = (qy > (float)(h - 1)) ? (float)(h - 1) : ((qy <= 0.0f) ? 0.0f : qy);
    // a 1-texel axis has no second sample: both taps read texel 0 (weights still sum to one)
    const int x_min = max(min((int)floorf(qx), w - 2), 0), x_max = min(x_min + 1, w - 1) > x_min ? x_min + 1 : x_min;
    const int y_min = max(min((int)floorf(qy), h - 2), 0), y_max = min(y_min + 1, h - 1) > y_min ? y_min + 1 : y_min;
    // the reference writes (x_max - q) / (x_max - x_min); the denominator is 1 whenever there are two taps, and a division
    // by 1.0f returns its numerator bit for bit
    f.wx0 = (x_max > x_min) ? ((float)x_max - qx) : 1.0f;
    f.wx1 = (x_max > x_min) ? (qx - (float)x_min) : 0.0f;
    f.wy0 = (y_max > y_min) ? ((float)y_max - qy) : 1.0f;
    f.wy1 = (y_max > y_min) ? (qy - (float)y_min) : 0.0f;
    f.i11 = y_min * w + x_min;
    f.i12 = y_min * w + x_max;
    f.i21 = y_max * w + x_min;
    f.i22 = y_max * w + x_max;
    return f;
  }
  /** the interpolation itself, in the reference's order: along x on both rows, then along y */
  MPPI_TEX_HD static inline float interpolate(const LinearFootprint& f, const float q11, const float q12, const float q21,
                                              const float q22)
  {
    const float lo = q11 * f.wx0 + q12 * f.wx1;
    const float hi = q21 * f.wx0 + q22 * f.wx1;
    return lo * f.wy0 + hi * f.wy1;
  }

  /** two_d_texture_helper.cu:151-245 (queryTextureCPU): bilinear interpolation or the nearest texel, clamp or border */
  MPPI_TEX_HD inline void queryTexture(const int index, const float* point, float* out) const
  {
    const TextureParams2D& p = textures_[index];
    if (p.filter_mode == FILTER_POINT)
    {
      const int w = p.width, h = p.height;
      float qx = point[0] * (float)w - 0.5f;
      float qy = point[1] * (float)h - 0.5f;
      qx = (qx == qx) ? qx : 0.0f;
      qy = (qy == qy) ? qy : 0.0f;
      bool border = false;
      if (p.address_mode[0] == ADDRESS_CLAMP)
        qx = (qx > (float)(w - 1)) ? (float)(w - 1) : ((qx <= 0.0f) ? 0.0f : qx);
      else
        border = border || (qx > (float)(w - 1)) || (qx <= 0.0f);
      if (p.address_mode[1] == ADDRESS_CLAMP)
        qy = (qy > (float)(h - 1)) ? (float)(h - 1) : ((qy <= 0.0f) ? 0.0f : qy);
      else
        border = border || (qy > (float)(h - 1)) || (qy <= 0.0f);
      const int idx = border ? 0 : (int)roundf(qy) * w + (int)roundf(qx);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int ch = 0; ch < NC; ch++)
      {
        // both values first, then the select: written as `border ? border_color[ch] : data[..]` the compiler selects between
        // the two ADDRESSES (one in the kernel-argument / private copy of this object, one in global memory) and issues a
        // single flat load, which faults for the private one (seen with NC = 4 on the terrain-normals map)
        float texel = p.data[(size_t)idx * NC + ch];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(texel));
#endif
        const float edge = p.border_color[ch];
        out[ch] = border ? edge : texel;
      }
      return;
    }
    const LinearFootprint f = linearFootprint(index, point);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int ch = 0; ch < NC; ch++)
    {
      const float v = interpolate(f, p.data[(size_t)f.i11 * NC + ch], p.data[(size_t)f.i12 * NC + ch],
                                  p.data[(size_t)f.i21 * NC + ch], p.data[(size_t)f.i22 * NC + ch]);
      out[ch] = f.border ? p.border_color[ch] : v;
    }
  }

  /**
   * N lookups at world poses whose loads are all in flight together: footprints first, then the 4 N NC loads, then the
   * interpolations.  One lane per rollout sees a full memory round trip per dependent load; a loop over
   * queryTextureAtWorldPose() pays it N times (the elevation-map RACER step: four wheels).  Same values as N single
   * calls.  world: [N][3], out: [N][NC].
   */
  template <int N>
  MPPI_TEX_HD inline void queryTextureAtWorldPoseBatch(const int index, const float (*world)[3], float* out) const
  {
    const TextureParams2D& p = textures_[index];
    if (p.filter_mode == FILTER_POINT)
    {
      for (int n = 0; n < N; n++)
        queryTextureAtWorldPose(index, world[n], out + n * NC);
      return;
    }
    LinearFootprint f[N];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int n = 0; n < N; n++)
    {
      float map[3], tex[3];
      worldPoseToMapPose(index, world[n], map);
      mapPoseToTexCoord(index, map, tex);
      f[n] = linearFootprint(index, tex);
    }
    float q[N][4 * NC];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int n = 0; n < N; n++)
      for (int ch = 0; ch < NC; ch++)
      {
        q[n][4 * ch + 0] = p.data[(size_t)f[n].i11 * NC + ch];
        q[n][4 * ch + 1] = p.data[(size_t)f[n].i12 * NC + ch];
        q[n][4 * ch + 2] = p.data[(size_t)f[n].i21 * NC + ch];
        q[n][4 * ch + 3] = p.data[(size_t)f[n].i22 * NC + ch];
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int n = 0; n < N; n++)
      for (int ch = 0; ch < NC; ch++)
      {
        const float v = interpolate(f[n], q[n][4 * ch + 0], q[n][4 * ch + 1], q[n][4 * ch + 2], q[n][4 * ch + 3]);
        out[n * NC + ch] = f[n].border ? p.border_color[ch] : v;
      }
  }

  /** texture_helper.cu:274-280 */
  MPPI_TEX_HD inline void queryTextureAtWorldPose(const int index, const float* input, float* out) const
  {
    float map[3], tex[3];
    worldPoseToMapPose(index, input, map);
    mapPoseToTexCoord(index, map, tex);
    queryTexture(index, tex, out);
  }

  /** texture_helper.cu:283-289 */
  MPPI_TEX_HD inline void queryTextureAtMapPose(const int index, const float* input, float* out) const
  {
    float tex[3];
    mapPoseToTexCoord(index, input, tex);
    queryTexture(index, tex, out);
  }

private:
  MPPI_TEX_HD static inline int min(int a, int b)
  {
    return a < b ? a : b;
  }
};
}  // namespace texture
}  // namespace mppi

#endif
