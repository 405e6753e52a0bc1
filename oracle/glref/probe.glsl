#version 450 core
// oracle/glref/probe.glsl -- TEST INFRASTRUCTURE.  Calls individual functions of the reference's common.glsl (its text is
// spliced in below at run time, from /root/reference) on vectors of inputs, so that the oracle can be pinned function
// by function.  Two vec4 in, two vec4 out per item; `mode` selects the function.
layout (local_size_x = 64) in;
#define USE_DDA
@COMMON@
layout(std430, binding = 5) buffer ProbeIn { vec4 inp[]; };
layout(std430, binding = 6) buffer ProbeOut { vec4 outp[]; };
uniform int mode;
uniform int n_items;

void main() {
    const uint i = gl_GlobalInvocationID.x;
    if (i >= uint(n_items)) return;
    const vec4 a = inp[2 * i], b = inp[2 * i + 1];
    vec4 r0 = vec4(0), r1 = vec4(0);
    if (mode == 0) {                 // tea(v0, v1, 32) and the first LCG draws from it
        uint s = tea(floatBitsToUint(a.x), floatBitsToUint(a.y), 32);
        r0.x = uintBitsToFloat(s);
        r0.y = rng(s); r0.z = rng(s); r0.w = rng(s);
        r1.x = uintBitsToFloat(s);
    } else if (mode == 1) {          // brick fetches
        r0.x = lookup_density_brick(a.xyz);
        r0.y = lookup_majorant(a.xyz, 0); r0.z = lookup_majorant(a.xyz, 1); r0.w = lookup_majorant(a.xyz, 2);
        r1.x = lookup_majorant(a.xyz, 3);
    } else if (mode == 2) {          // environment importance sampling
        vec3 w_i;
        const vec4 e = sample_environment(a.xy, w_i);
        r0 = e; r1.xyz = w_i;
    } else if (mode == 3) {          // phase function and its sampler
        r0.x = phase_henyey_greenstein(a.w, b.x);
        r1.xyz = sample_phase_henyey_greenstein(a.xyz, b.x, b.yz);
    } else if (mode == 4) {          // camera ray
        r0.xyz = view_dir(ivec2(a.xy), ivec2(a.zw), b.xy);
    } else if (mode == 5) {          // box clip
        vec2 nf = vec2(0);
        r0.x = intersect_box(a.xyz, b.xyz, vol_bb_min, vol_bb_max, nf) ? 1.f : 0.f;
        r0.yz = nf;
    } else if (mode == 6) {          // one shadow segment (transmittanceDDA) with its RNG stream
        uint s = floatBitsToUint(a.w);
        r0.x = transmittanceDDA(a.xyz, b.xyz, s);
        r0.y = uintBitsToFloat(s);
    } else if (mode == 7) {          // built-ins the shaders rely on, as this GL implementation evaluates them
        r0 = vec4(log(a.x), sin(a.y), cos(a.y), acos(a.z));
        r1 = vec4(atan(a.w, b.x), exp(b.y), pow(b.z, b.w), sqrt(a.x));
    } else if (mode == 8) {          // bilinear environment fetch and the escape-path lookup
        r0.xyz = texture(env_envmap, a.xy).rgb;
        r1.xyz = lookup_environment(b.xyz);
        r1.w = pdf_environment(b.xyz);
    } else if (mode == 9) {          // one camera segment (sample_volumeDDA)
        uint s = floatBitsToUint(a.w);
        float t = 0.f; vec3 thr = vec3(1), Le = vec3(0);
        r0.x = sample_volumeDDA(a.xyz, b.xyz, t, thr, Le, s) ? 1.f : 0.f;
        r0.y = t; r0.z = uintBitsToFloat(s);
        r1.xyz = thr;
    }
    else if (mode == 10) {         // is fma() fused here?  is a*b+c contracted?
        r0.x = fma(a.x, a.y, a.z); r0.y = a.x * a.y + a.z;
        r0.z = a.x / a.y; r0.w = sqrt(abs(a.x));
        r1.x = inversesqrt(abs(a.x)); r1.yzw = normalize(a.xyz);
    }
    outp[2 * i] = r0; outp[2 * i + 1] = r1;
}
