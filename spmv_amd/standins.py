"""Synthetic stand-ins for the reference's SuiteSparse sweep list.

The reference benchmarks a list of 159 SuiteSparse matrices
(reference matrix_lists/matrix_list_160.txt:1-159, run by spmv.run:20-33).
Those files are not available offline, so every entry of the list gets a
generated matrix of the same family (structured grid, block FEM, scalar FEM,
circuit, KKT / saddle point, quantum chemistry, random) with about the same
number of rows and non-zeros.  The (rows, nnz) figures below are the published
SuiteSparse sizes quoted from memory (nnz after symmetric expansion where the
file is symmetric, which is what the reference's loader produces,
src/mtx.c:127-132); they are approximate by intent -- the stand-ins reproduce
shape classes, not the matrices.

standin(name, scale=...) returns (rows, cols, row_ptr, col, val) like the
generators module; everything is counter-based hashing on torch tensors, so the
same matrix comes out on any device.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

from .generators import _finish, _splitmix64, _stencil, _u01

# name, family, rows, nnz
TABLE = [
    ("G2_circuit", "grid5", 150102, 726674), ("G3_circuit", "grid5", 1585478, 7660826),
    ("pre2", "circuit", 659033, 5834044), ("twotone", "circuit", 120750, 1206265),
    ("BenElechi1", "fem", 245874, 13150496), ("CurlCurl_1", "fem1", 226451, 2472071),
    ("CurlCurl_2", "fem1", 806529, 8921789), ("CurlCurl_3", "fem1", 1219574, 13544618),
    ("CurlCurl_4", "fem1", 2380515, 26515867), ("pwtk", "fem", 217918, 11524432),
    ("FEM_3D_thermal2", "fem1", 147900, 3489300), ("thermomech_dM", "fem1", 204316, 1423116),
    ("thermomech_TC", "fem1", 102158, 711558), ("atmosmodd", "grid7", 1270432, 8814880),
    ("atmosmodl", "grid7", 1489752, 10319760), ("t2em", "grid5", 921632, 4590832),
    ("tmt_sym", "grid5", 726713, 5080961), ("tmt_unsym", "grid5", 917825, 4584801),
    ("ship_003", "fem", 121728, 8086034), ("shipsec1", "fem", 140874, 7813404),
    ("shipsec5", "fem", 179860, 10113096), ("shipsec8", "fem", 114919, 6653399),
    ("x104", "fem", 108384, 10167624), ("dielFilterV2real", "fem1", 1157456, 48538952),
    ("dielFilterV3real", "fem1", 1102824, 89306020), ("gsm_106857", "fem1", 589446, 21758924),
    ("HV15R", "fem", 2017169, 283073458), ("PR02R", "fem", 161070, 8185136),
    ("RM07R", "fem", 381689, 37464962), ("circuit5M_dc", "circuit", 3523317, 14865409),
    ("circuit5M", "circuit", 5558326, 59524291), ("Freescale1", "circuit", 3428755, 17052626),
    ("Freescale2", "circuit", 2999349, 14313235), ("FullChip", "circuit", 2987012, 26621983),
    ("memchip", "circuit", 2707524, 13343948), ("nxp1", "circuit", 414604, 2655880),
    ("transient", "circuit", 178866, 961368), ("bmw3_2", "fem", 227362, 11288630),
    ("boyd2", "kkt", 466316, 1500397), ("cont-300", "kkt", 180895, 988195),
    ("darcy003", "kkt", 389874, 2097566), ("d_pretok", "fem1", 182730, 1641672),
    ("helm2d03", "fem1", 392257, 2741935), ("mario002", "kkt", 389874, 2097566),
    ("turon_m", "fem1", 189924, 1690876), ("apache2", "grid7", 715176, 4817870),
    ("audikw_1", "fem", 943695, 77651847), ("bmw7st_1", "fem", 141347, 7318399),
    ("bmwcra_1", "fem", 148770, 10641602), ("hood", "fem", 220542, 9895422),
    ("inline_1", "fem", 503712, 36816170), ("ldoor", "fem", 952203, 42493817),
    ("Goodwin_095", "fem1", 100037, 3226066), ("Goodwin_127", "fem1", 178437, 5778545),
    ("TEM152078", "fem1", 152078, 6459326), ("TEM181302", "fem1", 181302, 7839010),
    ("hcircuit", "circuit", 105676, 513072), ("scircuit", "circuit", 170998, 958936),
    ("Hamrle3", "circuit", 1447360, 5514242), ("hvdc2", "circuit", 189860, 1339638),
    ("dc1", "circuit", 116835, 766396), ("trans4", "circuit", 116835, 749800),
    ("msdoor", "fem", 415863, 19173163), ("HTC_336_4438", "circuit", 226340, 783748),
    ("Bump_2911", "fem", 2911419, 127729899), ("CoupCons3D", "fem", 416800, 17277420),
    ("Cube_Coup_dt0", "fem", 2164760, 124406070), ("Emilia_923", "fem", 923136, 40373538),
    ("Fault_639", "fem", 638802, 27245944), ("Flan_1565", "fem", 1564794, 114165372),
    ("Geo_1438", "fem", 1437960, 60236322), ("Hook_1498", "fem", 1498023, 59374451),
    ("Long_Coup_dt0", "fem", 1470152, 84422970), ("ML_Geer", "fem", 1504002, 110686677),
    ("ML_Laplace", "fem", 377002, 27582698), ("PFlow_742", "fem", 742793, 37138461),
    ("Queen_4147", "fem", 4147110, 316548962), ("Serena", "fem", 1391349, 64131971),
    ("StocF-1465", "fem1", 1465137, 21005389), ("Transport", "fem1", 1602111, 23487281),
    ("ch7-8-b5", "random", 141120, 846720), ("m133-b3", "random", 200200, 800800),
    ("shar_te2-b3", "random", 200200, 800800), ("F1", "fem", 343791, 26837113),
    ("Lin", "grid7", 256000, 1766400), ("bundle_adj", "kkt", 513351, 20207907),
    ("ecology1", "grid5", 1000000, 4996000), ("lung2", "fem1", 109460, 492564),
    ("stomach", "fem1", 213360, 3021648), ("torso1", "fem1", 116158, 8516500),
    ("torso2", "fem1", 115967, 1033473), ("torso3", "fem1", 259156, 4429042),
    ("boneS01", "fem", 127224, 5516602), ("boneS10", "fem", 914898, 40878708),
    ("filter3D", "fem1", 106437, 2707179), ("CO", "qchem", 221119, 7666057),
    ("Ga10As10H30", "qchem", 113081, 6115633), ("Ga19As19H42", "qchem", 133123, 8884839),
    ("Ga41As41H72", "qchem", 268096, 18488476), ("Ge87H76", "qchem", 112985, 7892195),
    ("Ge99H100", "qchem", 112985, 8451395), ("Si41Ge41H72", "qchem", 185639, 15011265),
    ("Si87H76", "qchem", 240369, 10661631), ("SiO2", "qchem", 155331, 11283503),
    ("power197k", "circuit", 197156, 3568176), ("analytics", "random", 303813, 2006126),
    ("crashbasis", "kkt", 160000, 1750416), ("largebasis", "kkt", 440020, 5240084),
    ("majorbasis", "kkt", 160000, 1750416), ("Raj1", "circuit", 263743, 1300261),
    ("rajat21", "circuit", 411676, 1876011), ("rajat23", "circuit", 110355, 555441),
    ("rajat24", "circuit", 358172, 1946979), ("rajat29", "circuit", 643994, 3760246),
    ("rajat30", "circuit", 643994, 6175244), ("rajat31", "circuit", 4690002, 20316253),
    ("xenon2", "fem1", 157464, 3866688), ("cfd2", "fem1", 123440, 3085406),
    ("ASIC_320k", "circuit", 321821, 1931828), ("ASIC_680k", "circuit", 682862, 2638997),
    ("af_0_k101", "fem", 503625, 17550675), ("af_shell10", "fem", 1508065, 52259885),
    ("af_shell1", "fem", 504855, 17562051), ("c-73", "kkt", 169422, 1279274),
    ("c-big", "kkt", 345241, 2340859), ("matrix_9", "fem1", 103430, 1205518),
    ("matrix-new_3", "fem1", 125329, 893984), ("barrier2-1", "fem1", 113076, 2129496),
    ("ohne2", "fem1", 181343, 6869939), ("para-4", "fem1", 153226, 2930882),
    ("nlpkkt120", "kkt", 3542400, 95117792), ("nlpkkt160", "kkt", 8345600, 225422112),
    ("nlpkkt200", "kkt", 16240000, 440225632), ("nlpkkt240", "kkt", 27993600, 760648352),
    ("nlpkkt80", "kkt", 1062400, 28192672), ("thermal2", "fem1", 1228045, 8580313),
    ("engine", "fem", 143571, 4706073), ("TSOPF_FS_b39_c30", "circuit", 120216, 3121160),
    ("2cubes_sphere", "fem1", 101492, 1647264), ("offshore", "fem1", 259789, 4242673),
    ("Dubcova3", "fem1", 146689, 3636643), ("dgreen", "fem1", 1200611, 26606169),
    ("imagesensor", "fem1", 118758, 1446396), ("nv2", "fem1", 1453908, 37475646),
    ("power9", "circuit", 155376, 1887730), ("radiation", "fem1", 223104, 7637688),
    ("ss1", "fem1", 205282, 845089), ("ss", "fem1", 1652680, 34753577),
    ("stokes", "kkt", 11449533, 349321980), ("test1", "fem1", 392908, 9447535),
    ("vas_stokes_1M", "kkt", 1090664, 34767207), ("vas_stokes_2M", "kkt", 2146677, 65129037),
    ("vas_stokes_4M", "kkt", 4382246, 131577616), ("Baumann", "grid7", 112211, 748331),
    ("cop20k_A", "fem1", 121192, 2624331), ("mac_econ_fwd500", "random", 206500, 1273389),
    ("mc2depi", "grid5", 525825, 2100225), ("parabolic_fem", "grid7", 525825, 3674625),
    ("kkt_power", "kkt", 2063494, 12771361),
]

FAMILIES = ("grid5", "grid7", "fem", "fem1", "circuit", "kkt", "qchem", "random")
BY_NAME = {t[0]: t for t in TABLE}


def names():
    return [t[0] for t in TABLE]


def _hash(a: torch.Tensor, salt: int) -> torch.Tensor:
    return _u01(_splitmix64(a * 0x2545F491 + salt))


def _grid(family, rows, device):
    if family == "grid5":
        n = max(2, int(round(rows ** 0.5)))
        return _stencil((n, n), [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)], 4.0, -1.0, 0, n * n, device)
    n = max(2, int(round(rows ** (1.0 / 3.0))))
    offs = [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0)]
    return _stencil((n, n, n), offs, 6.0, -1.0, 0, n ** 3, device)


def _node_lengths(family, nn, mean, i, seed):
    """Neighbour count per node (>= 1: the diagonal), mean about `mean`."""
    u = _hash(i, seed + 1)
    if family in ("fem", "fem1", "kkt"):
        m = mean * (0.75 + 0.5 * u)
    elif family == "qchem":
        m = mean * (0.5 + u)
    elif family == "circuit":                      # most rows short, a heavy tail of long ones
        s = max(mean - 1.0, 0.1) / 2.33
        m = 1.0 + s * (u.clamp(min=1e-9) ** -0.7 - 1.0)
        m = m.clamp(max=max(nn / 16.0, 2.0))
    else:                                          # random: fixed length
        m = torch.full_like(u, mean)
    return m.round().to(torch.int64).clamp(min=1, max=nn)


def _node_columns(family, nn, mean, row, k, seed):
    """Column (node index) of the k-th neighbour of `row`; k == 0 is the diagonal."""
    e = (row << 20) ^ k
    u1, u2, u3 = _hash(e, seed + 2), _hash(e, seed + 3), _hash(e, seed + 4)
    sym = u1 * 2.0 - 1.0
    local = max(8.0, 2.0 * mean)
    if family in ("fem", "fem1"):
        far = max(local, float(nn) ** (2.0 / 3.0))
        off = torch.where(u2 < 0.75, sym * local, sym * far)
        c = row + off.to(torch.int64)
    elif family == "kkt":                          # mesh-like blocks coupled at a distance of 2/3 of the rows
        shift = (2 * nn) // 3
        c = row + (sym * local).to(torch.int64) + torch.where(u2 < 0.6, 0, shift)
    elif family == "circuit":
        anywhere = (u1 * nn).to(torch.int64)
        hub = (u1 * 64).to(torch.int64)
        c = torch.where(u2 < 0.7, row + (sym * 32).to(torch.int64), torch.where(u2 < 0.75, hub, anywhere))
    elif family == "qchem":
        dist = torch.exp(u3 * float(np.log(max(nn, 2)))).to(torch.int64)
        c = torch.where(u2 < 0.5, row + (sym * local).to(torch.int64), row + torch.where(u1 < 0.5, -dist, dist))
    else:
        c = (u1 * nn).to(torch.int64)
    c = torch.remainder(c, nn)
    return torch.where(k == 0, row, c)


def standin(name: str, scale: float = 1.0, device="cpu"):
    """The stand-in for list entry `name`, with rows and nnz multiplied by `scale`."""
    _, family, rows, nnz = BY_NAME[name]
    return generate(family, max(int(rows * scale), 64), max(int(nnz * scale), 64), zlib.crc32(name.encode()), device)


def generate(family: str, rows: int, nnz: int, seed: int = 0, device="cpu"):
    """A matrix of `family` with about `rows` rows and `nnz` non-zeros (CSR, columns ascending)."""
    if family not in FAMILIES:
        raise ValueError(f"unknown family {family!r}")
    if family in ("grid5", "grid7"):
        return _grid(family, rows, device)
    b = 3 if family == "fem" else 1                # dense b x b blocks per node pair
    nn = max(rows // b, 2)
    # duplicate neighbours are dropped below; draw a few more to land near the target
    oversample = {"circuit": 1.3, "random": 1.03}.get(family, 1.09)
    mean = max(nnz / float(nn * b * b) * oversample, 1.0)
    i = torch.arange(nn, dtype=torch.int64, device=device)
    lens = _node_lengths(family, nn, mean, i, seed)
    start = torch.cumsum(lens, 0) - lens
    row = torch.repeat_interleave(i, lens)
    k = torch.arange(int(lens.sum()), dtype=torch.int64, device=device) - torch.repeat_interleave(start, lens)
    del start, lens
    c = _node_columns(family, nn, mean, row, k, seed)
    del k
    keys = torch.unique((row << 32) | c)
    del row, c
    if b > 1:
        r, c = keys >> 32, keys & 0xFFFFFFFF
        keys = torch.cat([((b * r + d1) << 32) | (b * c + d2) for d1 in range(b) for d2 in range(b)])
        del r, c
        keys = torch.sort(keys).values
    n = nn * b
    row = keys >> 32
    col = keys & 0xFFFFFFFF
    val = _u01(_splitmix64(keys ^ seed))
    val = torch.where(row == col, val + 4.0, val)
    counts = torch.bincount(row, minlength=n)
    return _finish(n, n, counts, col, val)
