"""Scene mode, host side: tile grid, names and sharding follow reference src/load_data/tile_tifs.py:13-47
and the shipped tile names (src/utils.py:372-389)."""
import os

import numpy as np
import pytest

from aquaculture_amd import scenes


def _reference_loop(width, height, tilesize=1024):
    """Literal restatement of the loop nest of tile_tifs.py:33-36 (what gdal.Translate gets as srcWin)."""
    out = []
    for i in range(0, width, tilesize):
        for j in range(0, height, tilesize):
            w = min(i + tilesize, width) - i
            h = min(j + tilesize, height) - j
            out.append((i, j, w, h))
    return out


@pytest.mark.parametrize("wh", [(6144, 6144), (2500, 2100), (1024, 1024), (1023, 5), (1025, 2049), (12288, 6144)])
def test_tile_grid_is_the_reference_loop(wh):
    g = scenes.tile_grid(*wh)
    assert g == _reference_loop(*wh)
    assert sum(w * h for _, _, w, h in g) == wh[0] * wh[1]            # tiles partition the raster
    if wh == (6144, 6144):
        assert len(g) == 36 and g[1] == (0, 1024, 1024, 1024) and g[6] == (1024, 0, 1024, 1024)   # x outer, y inner


def test_bad_grid_arguments_are_refused():
    for bad in [(0, 10, 1024), (10, -1, 1024), (10, 10, 0)]:
        with pytest.raises(ValueError):
            scenes.tile_grid(*bad)


def test_tile_names_parse_like_the_shipped_ones():
    """The consumer splits the tile stem at '_' into exactly four fields (src/utils.py:385-389); year = last 4 chars of the first."""
    from aquaculture_amd import geocode
    for scene, year in [("/x/ORTHOIMAGERY.ORTHOPHOTOS2015_412.tif", "2015"), ("ORTHOIMAGERY.ORTHOPHOTOS.ORTHO-EXPRESS.2021_7.tiff", "2021")]:
        stem = scenes.tile_stem(scene, 2048, 5120)
        name, bbox_ind, x_off, y_off = stem.split("_")
        assert (name[-4:], x_off, y_off) == (year, "2048", "5120")
        ind, xo, yo, yr = geocode.parse_stems([stem])
        assert (int(ind[0]), int(xo[0]), int(yo[0]), int(yr[0])) == (int(bbox_ind), 2048, 5120, int(year))


def _write_scene(path, w, h, seed=0, mode="RGB"):
    from PIL import Image
    rng = np.random.default_rng(seed)
    ch = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
    arr = rng.integers(0, 256, (h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
    Image.fromarray(arr, mode).save(path, compression="tiff_lzw" if seed % 2 else None)
    return arr


def test_read_scene_and_rejections(tmp_path):
    a = _write_scene(str(tmp_path / "s.tif"), 300, 200, 1)
    got = scenes.read_scene(str(tmp_path / "s.tif"))
    assert got.dtype == np.uint8 and got.flags.c_contiguous and np.array_equal(got, a)
    g = _write_scene(str(tmp_path / "g.tif"), 40, 30, 2, "L")
    assert np.array_equal(scenes.read_scene(str(tmp_path / "g.tif")), np.repeat(g[:, :, None], 3, 2))
    _write_scene(str(tmp_path / "a.tif"), 40, 30, 3, "RGBA")
    with pytest.raises(ValueError):
        scenes.read_scene(str(tmp_path / "a.tif"))
    with pytest.raises(FileNotFoundError):
        scenes.list_scenes(str(tmp_path / "nothing"))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        scenes.list_scenes(str(tmp_path / "empty"))


def test_scene_shards_cover_every_tile_once(tmp_path):
    sizes = [(2100, 1100), (1024, 1024), (500, 2050), (1030, 1030), (64, 64)]
    for k, (w, h) in enumerate(sizes):
        _write_scene(str(tmp_path / f"ORTHOIMAGERY.ORTHOPHOTOS2015_{k}.tif"), w, h, k)
    total = sum(len(scenes.tile_grid(w, h)) for w, h in sizes)
    seen, names = [], []
    for world in (1, 2, 3):
        seen_w, names_w = [], []
        for rank in range(world):
            st = scenes.SceneTiles(str(tmp_path), shard=(rank, world), workers=2, batch_size=3)
            assert st.total == total
            for path, arr, stems, origins, (h, w), gids in st.batches():
                assert len(stems) == len(origins) == len(gids) <= 3
                for (x0, y0), s in zip(origins, stems):
                    assert s == scenes.tile_stem(path, x0, y0) and y0 + h <= arr.shape[0] and x0 + w <= arr.shape[1]
                seen_w += gids
                names_w += stems
        assert sorted(seen_w) == list(range(total)) and len(set(names_w)) == total
        seen.append(sorted(zip(seen_w, names_w)))
    assert seen[0] == seen[1] == seen[2]               # global tile index <-> name does not depend on the world size


def test_group_tiles_keeps_grid_order():
    g = scenes.tile_grid(2500, 1500)
    groups = scenes.group_tiles(g)
    assert list(groups) == [(1024, 1024), (476, 1024), (1024, 452), (476, 452)]
    assert groups[(1024, 1024)] == [0, 2] and groups[(476, 452)] == [5]


def test_mapped_scene_reads_the_same_bytes(tmp_path):
    """Uncompressed strip TIFFs are memory-mapped (no decode); anything else falls back to the decoder: same pixels either way."""
    a = _write_scene(str(tmp_path / "plain.tif"), 700, 300, 4)           # even seed: uncompressed
    m = scenes.read_scene(str(tmp_path / "plain.tif"), mapped=True)
    assert isinstance(m, np.memmap) and m.shape == (300, 700, 3) and np.array_equal(m, a)
    b = _write_scene(str(tmp_path / "lzw.tif"), 700, 300, 5)             # odd seed: LZW
    z = scenes.read_scene(str(tmp_path / "lzw.tif"), mapped=True)
    assert not isinstance(z, np.memmap) and np.array_equal(z, b)
    g = _write_scene(str(tmp_path / "grey.tif"), 64, 32, 6, "L")
    assert not isinstance(scenes.read_scene(str(tmp_path / "grey.tif"), mapped=True), np.memmap)
    # a truncated pixel block must not be mapped past the end of the file
    raw = open(tmp_path / "plain.tif", "rb").read()
    open(tmp_path / "cut.tif", "wb").write(raw[: len(raw) // 2])
    assert scenes._mapped_strips(str(tmp_path / "cut.tif")) is None


def test_tile_grid_properties_hold_for_arbitrary_sizes():
    """Property form of the grid contract (hypothesis): the tiles partition the raster exactly once, come in the reference's order
    (x outer, y inner), no tile is larger than the tile size and only the last row / column is cut short."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 9000), st.integers(1, 9000), st.sampled_from([256, 640, 1000, 1024]))
    def check(w, h, ts):
        g = scenes.tile_grid(w, h, ts)
        assert g == _reference_loop(w, h, ts)
        assert sum(tw * th for _, _, tw, th in g) == w * h and len(set((x, y) for x, y, _, _ in g)) == len(g)
        assert g == sorted(g, key=lambda t: (t[0], t[1]))
        for x, y, tw, th in g:
            assert 0 < tw <= ts and 0 < th <= ts and x % ts == 0 and y % ts == 0
            assert (tw == ts or x + tw == w) and (th == ts or y + th == h)
    check()
