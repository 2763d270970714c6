// svr_primary.hpp -- a path up to its first scatter event, shared by the tile kernel (svr_trace_tile.hip) and the split kernels of
// deeper paths (svr_trace_split.hip).
#pragma once
#include "svr_walk.hpp"
#include "svr_lanes.hpp"

namespace svr {

#ifndef SVR_SHADOW_REMARCH
#define SVR_SHADOW_REMARCH true
#endif
#ifndef SVR_PRIMARY_REMARCH
#define SVR_PRIMARY_REMARCH false
#endif

// The part of a path up to its first scatter event (k = 0 of kernel_pathtracer's loop, pathtracer.cu:205-235) for QUEUE
// builds: returns true if the primary walk collided -- pt / wo / val / rng are then the scatter event the wave shades in place and
// svr_lanes.hpp continues the path -- and false if the path is over, with its radiance in L.
template <int LAYOUT, bool COUNT, bool SKIP, typename LDS>
SVR_DEV bool trace_primary(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t hashed, bool group_march, uint32_t P2,
                           GroupMapShared* gslot, Cnt& c, Rng& rng, v3& L, v3& pt, v3& wo, float& val, const DevScene* scp = nullptr)
{
    const DevScene& sc = scp ? *scp : s;                // (set-up constants: camera, lights, environment -- svr_lanes.hpp, shade_event)
    uint32_t offset = y * sc.imageW + x;
    rng_init(rng, hashed + offset);
    if (COUNT) c.paths++;
    L = V3(0.f, 0.f, 0.f);
    const v3 T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(sc, x, y, rng, orig, dir);
    float ls_t;
    int ls_id = nearest_light(sc, orig, dir, ls_t);
    float tMin = (float)1e-6, tMax = SVR_FLT_MAX;
    val = 0.f;
    float t;
    if (SKIP && group_march) {
        float t_occ;
        GroupMap map;
        map.g = gslot + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
        int r = walk_setup_group<COUNT, SKIP>(s, L_, P2, orig, dir, false, tMin, tMax, t_occ, map);
        t = r <= 0 ? -SVR_FLT_MAX
                   : walk_run<LAYOUT, COUNT, SKIP, false, true>(s, L_, orig, dir, rng, tMin, tMax, t_occ, val, false, c, &map, P2);
    } else
        t = walk<LAYOUT, COUNT, SKIP, SVR_PRIMARY_REMARCH>(s, L_, orig, dir, rng, tMin, tMax, val, false, c);
    if (ls_id >= 0) {
        float tt = t < 0.f ? SVR_FLT_MAX : t;
        if (ls_t < tt) {
            const DevLight& l = sc.lights[ls_id];
            float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
            L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
            return false;
        }
    }
    if (t < 0.f) {
        if (sc.env_on_escape) L = L + T * env_radiance(sc, dir);
        return false;
    }
    wo = -dir;
    pt = orig + dir * t;
    return true;
}

// POOL builds: trace_primary up to the walk -- generator, camera ray, nearest light, box, whole-ray test.  true = the ray has a walk to do
// (orig / dir / tMin / tMax / t_occ / ls_* describe it: a P record, svr_lanes.hpp); false = the path is over with its radiance in L.
template <bool COUNT, bool SKIP, typename LDS>
SVR_DEV bool gen_primary(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t hashed, bool group_march, uint32_t P2, GroupMapShared* gslot,
                         Cnt& c, Rng& rng, v3& L, v3& orig, v3& dir, float& tMin, float& tMax, float& t_occ, float& ls_t, int& ls_id, const DevScene* scp = nullptr)
{
    const DevScene& sc = scp ? *scp : s;
    uint32_t offset = y * sc.imageW + x;
    rng_init(rng, hashed + offset);
    if (COUNT) c.paths++;
    L = V3(0.f, 0.f, 0.f);
    const v3 T = V3(1.f, 1.f, 1.f);
    camera_ray(sc, x, y, rng, orig, dir);
    ls_id = nearest_light(sc, orig, dir, ls_t);
    tMin = (float)1e-6; tMax = SVR_FLT_MAX;
    int r;
    if (SKIP && group_march) {
        GroupMap map;
        map.g = gslot + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
        r = walk_setup_group<COUNT, SKIP>(s, L_, P2, orig, dir, false, tMin, tMax, t_occ, map);
    } else
        r = walk_setup<COUNT, SKIP>(s, L_, orig, dir, false, tMin, tMax, t_occ);
    if (r > 0) return true;
    // no walk: the result of sample_distance is -FLT_MAX (pathtracer.cu:220-235 with t < 0)
    if (ls_id >= 0) {
        const DevLight& l = sc.lights[ls_id];
        float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
        L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
        return false;
    }
    if (sc.env_on_escape) L = L + T * env_radiance(sc, dir);
    return false;
}

} // namespace svr
