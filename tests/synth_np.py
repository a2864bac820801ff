"""numpy restatement of include/amt_synth.h (independent check of the input generator)."""
import numpy as np

FIELDS = ("ww", "ww_1", "u", "u_1", "v", "v_1", "mu", "mut", "muave", "muts", "muu", "muv",
          "mudf", "t", "t_1", "t_ave", "ft", "mu_tend", "dnw", "fnm", "fnp", "rdnw",
          "msfuy", "msfvx_inv", "msftx", "msfty")
RANK3 = {"ww", "ww_1", "u", "u_1", "v", "v_1", "t", "t_1", "t_ave", "ft"}
RANK1 = {"dnw", "fnm", "fnp", "rdnw"}
PARAMS = {  # base, smooth amplitude, noise amplitude
    "u": (10.0, 4.0, 0.1), "v": (-6.0, 3.0, 0.1), "u_1": (1.0e-4, 4.0e-5, 1.0e-5),
    "v_1": (-7.0e-5, 3.0e-5, 1.0e-5), "t": (300.0, 3.0, 0.5), "t_1": (299.0, 3.0, 0.5),
    "t_ave": (-777.0, 0.0, 1.0), "ft": (0.0, 3.0e-3, 1.0e-2), "ww": (0.0, 0.02, 0.002),
    "ww_1": (0.0, 0.015, 0.002), "mu": (10.0, 3.0, 2.0), "mut": (9.0e4, 600.0, 300.0),
    "muu": (9.0e4, 600.0, 300.0), "muv": (9.0e4, 600.0, 300.0), "mu_tend": (0.0, 3.0e-3, 1.0e-2),
    "muave": (-555.0, 0.0, 1.0), "muts": (-444.0, 0.0, 1.0), "mudf": (-333.0, 0.0, 1.0),
    "msfuy": (1.0, 0.06, 0.04), "msfvx_inv": (1.0, 0.06, 0.04), "msftx": (1.0, 0.06, 0.04),
    "msfty": (1.0, 0.06, 0.04),
}
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _u64(x):
    return np.asarray(x).astype(np.int64).astype(np.uint64)


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def tri(n, P):
    m = np.asarray(n, dtype=np.int64) & (P - 1)
    return np.abs(2 * m - P).astype(np.float64) / float(P)


def synth(name, seed, shape_ikj, g0, gdims, dtype=np.float64):
    """shape_ikj = (idim,kdim,jdim) of the patch, g0 = (gi0,gk0,gj0), gdims = (gidim,gkdim,gjdim).
    Returns an array shaped like the patch field (j,k,i) / (j,i) / (k,)."""
    f = FIELDS.index(name)
    idim, kdim, jdim = shape_ikj
    gi0, gk0, gj0 = g0
    gidim, gkdim, gjdim = gdims
    if name in RANK1:
        gk = np.arange(kdim, dtype=np.int64) + gk0
        nk = gkdim - 1 if gkdim > 1 else 1
        wob = tri(gk, 8) - 0.5
        dnw = -(1.0 / float(nk)) * (1.0 + 0.25 * wob)
        fnm = 0.5 + 0.125 * (tri(gk + 3, 16) - 0.5)
        out = {"dnw": dnw, "rdnw": 1.0 / dnw, "fnm": fnm, "fnp": 1.0 - fnm}[name]
        return out.astype(dtype)
    gi = (np.arange(idim, dtype=np.int64) + gi0)
    gj = (np.arange(jdim, dtype=np.int64) + gj0)
    with np.errstate(over="ignore"):
        if name in RANK3:
            gk = (np.arange(kdim, dtype=np.int64) + gk0)
            GJ, GK, GI = np.meshgrid(gj, gk, gi, indexing="ij")
            lin = (_u64(GJ) * np.uint64(gkdim) + _u64(GK)) * np.uint64(gidim) + _u64(GI)
            smooth = tri(GI + 5 * f, 64) + tri(GJ + 11 * f, 32) + tri(GK + 3 * f, 16) - 1.5
        else:
            GJ, GI = np.meshgrid(gj, gi, indexing="ij")
            lin = _u64(GJ) * np.uint64(gidim) + _u64(GI)
            smooth = tri(GI + 7 * f, 128) + tri(GJ + 13 * f, 64) - 1.0
        h = splitmix64(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
                       + np.uint64(f + 1) * np.uint64(0xD1B54A32D192ED03) + lin)
    r = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    s = 2.0 * r - 1.0
    base, a_s, a_n = PARAMS[name]
    return (base + a_s * smooth + a_n * s).astype(dtype)
