"""CPU: the XCD-aware tile placement of the persistent 3D kernel (cspn3d_persistent.hip, round 5).  The kernel derives its tile from
blockIdx exactly as `tile_of` below; the plan comes from the C++ planner through the hook library (no GPU needed)."""
import ctypes

import pytest

from cspn_amd import _lib


def geo3(B, D, H, W, N):
    info = (ctypes.c_int * 9)()
    _lib.load_hooks().cspn_debug_3d_geo(B, D, H, W, N, info)
    return dict(zip(("tz", "ty", "cx", "tiles", "launched", "bz", "by", "bx", "chunks"), list(info)))


def tile_of(b, g):
    """numpy-free twin of the kernel's prologue: workgroup id -> (iz, iy, ix) or None"""
    if g["bz"] > 0:
        nbx, nby = g["cx"] // g["bx"], g["ty"] // g["by"]
        k, s = b & 7, b >> 3
        kx, ky, kz = k % nbx, (k // nbx) % nby, k // (nbx * nby)
        sx, sy, sz = s % g["bx"], (s // g["bx"]) % g["by"], s // (g["bx"] * g["by"])
        iz, iy, ix = kz * g["bz"] + sz, ky * g["by"] + sy, kx * g["bx"] + sx
        return (iz, iy, ix) if s < g["bz"] * g["by"] * g["bx"] and iz < g["tz"] else None
    if b >= g["tiles"]:
        return None
    return (b // (g["cx"] * g["ty"]), (b // g["cx"]) % g["ty"], b % g["cx"])


@pytest.mark.parametrize("shape", [(4, 32, 160, 608, 12), (1, 32, 160, 152, 6), (2, 16, 64, 200, 5), (1, 64, 64, 128, 3), (1, 16, 24, 128, 6),
                                   (8, 32, 160, 608, 4), (1, 8, 8, 64, 2), (3, 9, 17, 72, 4), (2, 20, 30, 200, 12), (1, 40, 56, 64, 3)])
def test_every_tile_is_owned_by_exactly_one_workgroup_and_blocks_share_an_id_class(shape):
    g = geo3(*shape)
    assert g["tiles"] == g["tz"] * g["ty"] * g["cx"] and g["tiles"] <= g["launched"] <= 256
    owners = {}
    for b in range(g["launched"]):
        t = tile_of(b, g)
        if t is not None:
            assert 0 <= t[0] < g["tz"] and 0 <= t[1] < g["ty"] and 0 <= t[2] < g["cx"]
            assert t not in owners, (t, b, owners[t])
            owners[t] = b
    assert len(owners) == g["tiles"]
    if g["bz"] > 0:
        assert g["tz"] % g["bz"] == 0 and g["ty"] % g["by"] == 0 and g["cx"] % g["bx"] == 0
        blocks = (g["tz"] // g["bz"]) * (g["ty"] // g["by"]) * (g["cx"] // g["bx"])
        assert 4 <= blocks <= 8 and g["bz"] * g["by"] * g["bx"] <= 32
        # all tiles of a block have workgroup ids in one residue class mod 8 (ids 8 apart have so far shared an XCD)
        for (iz, iy, ix), b in owners.items():
            blk = ((iz // g["bz"]) * (g["ty"] // g["by"]) + iy // g["by"]) * (g["cx"] // g["bx"]) + ix // g["bx"]
            assert b % 8 == blk


def test_config5_plan_is_eight_blocks_of_thirty():
    g = geo3(4, 32, 160, 608, 12)
    assert (g["tz"], g["ty"], g["cx"], g["bz"], g["by"], g["bx"], g["launched"], g["chunks"]) == (4, 20, 3, 2, 5, 3, 240, 15)
    # share of a tile's fetched quads that cross a block face (z face 10 rows x 24 quads, y face 8 x 24, x face 100 singles)
    ext = tot = 0
    for iz in range(4):
        for iy in range(20):
            for ix in range(3):
                for dz, dy, dx, q in ((-1, 0, 0, 240), (1, 0, 0, 240), (0, -1, 0, 192), (0, 1, 0, 192), (0, 0, -1, 100), (0, 0, 1, 100)):
                    z, y, x = iz + dz, iy + dy, ix + dx
                    if 0 <= z < 4 and 0 <= y < 20 and 0 <= x < 3:
                        tot += q
                        ext += q * ((z // 2, y // 5, x // 3) != (iz // 2, iy // 5, ix // 3))
    assert ext / tot < 0.25     # (plain order: every neighbour on another XCD)
