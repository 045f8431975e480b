"""A SECOND, independent statement of march_rays_train (raymarching/src/raymarching.cu:314-490), written from the
kernel's description in scalar numpy float32 arithmetic -- not from oracle/enerf_oracle.c.
TEST INFRASTRUCTURE ONLY: tests/test_march_second_statement.py requires the C oracle to agree with it bit for bit, which
is what stands in for the golden vectors the reference does not have for its marcher (slow: pure Python loops; used on a
few dozen rays).

Floating-point conventions (the ones integer outputs depend on):
  * every `a*b + c` the CUDA compiler contracts is a single-rounded fma here (computed exactly in float64 -- a product
    of two float32 is exact in float64 -- and rounded once);
  * the cell coordinate is `(float)(0.5 * (double)fmaf(x, 1/mip_bound, 1) * (double)H)` (the literal 0.5 is a double);
  * the voxel-exit parameter uses (H - 1) where the cell coordinate used H (kept as in the reference).
"""
import math

import numpy as np

F = np.float32
SQRT3 = F(1.7320508075688772)


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def clamp(v, lo, hi):
    return F(min(F(hi), max(F(lo), F(v))))


def mip_exponent(v, C):
    e = math.frexp(float(v))[1]                     # |v| in [2^(e-1), 2^e), e = 0 for v == 0
    return int(min(C - 1, max(0, e)))


def morton3(x, y, z):
    out = 0
    for b in range(10):
        out |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return out


def pcg32_first_float(seed, seq):
    M, mask = 0x5851f42d4c957f2d, (1 << 64) - 1
    state, inc = 0, ((seq << 1) | 1) & mask

    def nxt():
        nonlocal state
        old = state
        state = (old * M + inc) & mask
        xs = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xffffffff
    nxt()
    state = (state + seed) & mask
    nxt()
    u = (nxt() >> 9) | 0x3f800000
    return F(np.array([u], np.uint32).view(np.float32)[0] - F(1.0))


def march_one(o, d, grid_bits, bound, dt_gamma, max_steps, C, H, near, far, perturb, n, limit, emit, seq=1):
    """One ray.  emit(x, y, z, dt, t_after) is called per occupied step; returns the number of steps taken.
    Jitter: t0 = near + dt_min * pcg32(n, seq).next_float() when `perturb` (A.4)."""
    ox, oy, oz = (F(v) for v in o)
    dx, dy, dz = (F(v) for v in d)
    with np.errstate(divide="ignore"):
        rdx, rdy, rdz = F(1) / dx, F(1) / dy, F(1) / dz
    bound, dt_gamma = F(bound), F(dt_gamma)
    dt_min = F(F(2) * SQRT3 / F(max_steps))
    dt_max = F(F(2) * SQRT3 * F(1 << (C - 1)) / F(H))
    t = F(near)
    if perturb:
        t = fma(dt_min, pcg32_first_float(n, seq), t)
    steps = 0
    hm1 = F(H - 1)
    while t < F(far) and steps < limit:
        x = clamp(fma(t, dx, ox), -bound, bound)
        y = clamp(fma(t, dy, oy), -bound, bound)
        z = clamp(fma(t, dz, oz), -bound, bound)
        dt = clamp(F(t * dt_gamma), dt_min, dt_max)
        level = max(mip_exponent(max(abs(x), abs(y), abs(z)), C), mip_exponent(F(np.float64(F(dt * F(H))) * 0.5), C))
        mip_bound = F(min(F(1 << level), bound))
        mip_rbound = F(F(1) / mip_bound)

        def cell(v):
            c = F(0.5 * np.float64(fma(v, mip_rbound, F(1))) * np.float64(H))
            return int(clamp(c, F(0), hm1))
        nx, ny, nz = cell(x), cell(y), cell(z)
        index = level * H * H * H + morton3(nx, ny, nz)
        if grid_bits[index >> 3] & (1 << (index & 7)):
            t = F(t + dt)
            emit(x, y, z, dt, t)
            steps += 1
        else:
            def exit_t(nc, dc, pc, rdc):
                sgn = F(math.copysign(1.0, float(dc)))
                face = fma(F(F(F(nc) + F(0.5) + F(F(0.5) * sgn)) / hm1), F(2), F(-1))
                with np.errstate(invalid="ignore", over="ignore"):
                    return F(fma(face, mip_bound, -pc) * rdc)
            with np.errstate(invalid="ignore"):
                tx, ty, tz = exit_t(nx, dx, x, rdx), exit_t(ny, dy, y, rdy), exit_t(nz, dz, z, rdz)
                # fminf / fmaxf semantics: a NaN operand is ignored
                m = F(np.fmin(tx, np.fmin(ty, tz)))
                tt = F(t + F(np.fmax(F(0), m)))
            while True:
                t = F(t + clamp(F(t * dt_gamma), dt_min, dt_max))
                if not t < tt:
                    break
    return steps


def march_rays_train(rays_o, rays_d, grid_bits, bound, dt_gamma, max_steps, C, H, M, nears, fars, perturb):
    """-> xyzs [M,3], dirs [M,3], deltas [M,2] (zero where nothing is written), rays [N,3] int32, counter [2] int32, in
    sequential-execution order (ray n reserves after ray n-1: A.5)."""
    N = len(rays_o)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32)
    for n in range(N):
        args = (rays_o[n], rays_d[n], grid_bits, bound, dt_gamma, max_steps, C, H, nears[n], fars[n], perturb, n)
        num = march_one(*args, max_steps, lambda *a: None)                        # pass 1: count
        point_index, ray_index = int(counter[0]), int(counter[1])
        counter[0] += num
        counter[1] += 1
        rays[ray_index] = (n, point_index, num)
        if num == 0 or point_index + num >= M:                                     # note >=
            continue
        row = [point_index]
        state = {"last": None}

        def emit(x, y, z, dt, t_after, row=row, n=n, state=state):
            k = row[0]
            xyzs[k] = (x, y, z)
            dirs[k] = rays_d[n]
            if state["last"] is None:
                # deltas[1] = t_after - (t before the first emitted step's ... ) : the reference keeps `last_t`, which
                # starts at the ray's (jittered) t0
                state["last"] = state["t0"]
            deltas[k] = (dt, F(t_after - state["last"]))
            state["last"] = t_after
            row[0] = k + 1
        dt_min = F(F(2) * SQRT3 / F(max_steps))
        t0 = F(nears[n])
        if perturb:
            t0 = fma(dt_min, pcg32_first_float(n, 1), t0)
        state["t0"] = t0
        march_one(*args, num, emit)                                                # pass 2: write
    return xyzs, dirs, deltas, rays, counter


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, grid_bits, bound, dt_gamma, max_steps, C, H, fars, M,
               perturb):
    """Inference marching (raymarching.cu:701-813, A.4): alive slot n continues ray rays_alive[n] from rays_t[n] for at
    most n_step occupied samples, written at rows n*n_step..; unfilled rows stay zero.  The jitter seed is (slot n,
    perturb) and is added to the current t."""
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    dt_min = F(F(2) * SQRT3 / F(max_steps))
    for n in range(n_alive):
        idx = int(rays_alive[n])
        t0 = F(rays_t[n])
        if perturb:
            t0 = fma(dt_min, pcg32_first_float(n, perturb), t0)
        row = [n * n_step]
        last = [t0]

        def emit(x, y, z, dt, t_after, row=row, last=last, idx=idx):
            k = row[0]
            xyzs[k] = (x, y, z)
            dirs[k] = rays_d[idx]
            deltas[k] = (dt, F(t_after - last[0]))
            last[0] = t_after
            row[0] = k + 1
        march_one(rays_o[idx], rays_d[idx], grid_bits, bound, dt_gamma, max_steps, C, H, rays_t[n], fars[idx],
                  perturb, n, n_step, emit, seq=perturb if perturb else 1)
    return xyzs, dirs, deltas
