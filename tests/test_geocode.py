"""SURVEY.md 8f rank 3 (the consumer's label -> pixel box -> geocoded box arithmetic): oracle vs the reference's own output
(tests/golden/g7_geocode.json), oracle vs published EPSG worked examples, vectorised batch op vs oracle.  CPU only."""
import json
import os

import numpy as np
import pytest

from aquaculture_amd import geocode, postprocess
from oracle import geocode_oracle as GO

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "g7_geocode.json")) as f:
        g = json.load(f)
    g["bboxes"] = {int(k): tuple(v) for k, v in g["wanted_bboxes"].items()}
    return g


def test_oracle_pixel_to_3857_matches_reference_output(golden):
    """reference output/humanlabels.geojson polygons = convert_pix_to_m_bboxes of their pixel boxes: bit-exact but for 1-ulp cases."""
    exact, worst = 0, 0.0
    for ft in golden["features"]:
        label = ft["image"].replace(".jpeg", ".txt")
        x0, y0, x1, y1 = ft["pix"]
        xa, ya = GO.convert_pix_to_m_bboxes(x0, y0, label, golden["bboxes"])
        xb, yb = GO.convert_pix_to_m_bboxes(x1, y1, label, golden["bboxes"])
        got = [min(xa, xb), min(ya, yb), max(xa, xb), max(ya, yb)]
        err = max(abs(a - b) for a, b in zip(got, ft["poly_bounds_3857"]))
        worst = max(worst, err)
        exact += err == 0.0
        assert int(ft["image"].split("_")[0][-4:]) == ft["year"]
    assert worst <= 2.5e-10, worst                            # one ulp at 5e6 m
    assert exact >= 0.99 * len(golden["features"]), (exact, len(golden["features"]))


def test_oracle_3857_to_4326_matches_reference_im_center(golden):
    """`im_center` in the reference's output is the tile centre (pixel 512, 512) in EPSG:4326."""
    worst = 0.0
    for ft in golden["features"]:
        x, y = GO.convert_pix_to_m_bboxes(512, 512, ft["image"].replace(".jpeg", ".txt"), golden["bboxes"])
        lon, lat = GO.mercator_to_lonlat(x, y)
        worst = max(worst, abs(lat - ft["im_center_latlon"][0]), abs(lon - ft["im_center_latlon"][1]))
    assert worst < 1e-8, worst


def test_oracle_projection_known_answers():
    """IOGP Guidance Note 7-2 worked examples: LAEA Europe (EPSG:3035) and Popular Visualisation Pseudo-Mercator (EPSG:3857)."""
    e, n = GO.lonlat_to_laea_europe(5.0, 50.0)
    assert abs(e - 3962799.45) < 0.01 and abs(n - 2999718.85) < 0.01, (e, n)
    e0, n0 = GO.lonlat_to_laea_europe(10.0, 52.0)            # the projection origin maps to the false origin
    assert abs(e0 - 4321000.0) < 1e-6 and abs(n0 - 3210000.0) < 1e-6
    lon, lat = GO.mercator_to_lonlat(-11169055.58, 2800000.00)
    assert abs(lon - -(100 + 20 / 60)) < 1e-7 and abs(lat - (24 + 22 / 60 + 54.433 / 3600)) < 1e-7, (lon, lat)


def _random_labels(golden, n_images, seed):
    rng = np.random.default_rng(seed)
    names = sorted({ft["image"][:-5] for ft in golden["features"]})
    stems = [names[i] for i in rng.integers(0, len(names), n_images)]
    counts = rng.integers(1, 40, n_images)
    rows = np.zeros((int(counts.sum()), 6), np.float32)
    rows[:, 0] = rng.integers(0, 5, rows.shape[0])
    rows[:, 1:3] = rng.uniform(-0.02, 1.02, (rows.shape[0], 2))     # centres slightly outside the tile: negative pixels exercise trunc
    rows[:, 3:5] = rng.uniform(0.001, 0.3, (rows.shape[0], 2))
    rows[:, 5] = rng.uniform(0.25, 1.0, rows.shape[0])
    return stems, counts, rows


def test_batch_matches_oracle(golden):
    stems, counts, rows32 = _random_labels(golden, 60, 7)
    # the reference reads the %g text of the label files: go through the writer's text
    chunks, o = [], 0
    for c in counts:
        chunks.append(geocode.rows_from_label_text(postprocess.format_rows(rows32[o:o + c], True)))
        o += c
    rows = np.concatenate(chunks, 0)
    t = geocode.geocode_detections(stems, counts, rows, golden["bboxes"])
    k = 0
    for i, (s, c) in enumerate(zip(stems, counts)):
        ref = GO.geocode_label_rows(s + ".txt", rows[k:k + c], golden["bboxes"])
        for j, r in enumerate(ref):
            q = k + j
            assert (t["xmin"][q], t["xmax"][q], t["ymin"][q], t["ymax"][q]) == (r["xmin"], r["xmax"], r["ymin"], r["ymax"])
            assert (t["xmin_3857"][q], t["ymin_3857"][q], t["xmax_3857"][q], t["ymax_3857"][q]) == r["geometry"]     # bit-exact
            assert t["image"][q] == i and t["year"][q] == r["year"] and t["det_conf"][q] == r["det_conf"]
            assert geocode.REVERSE_CLASS_MAPPING[int(t["cls"][q])] == r["type"]
            for a, b in (("e_min_3035", "e_min_3035"), ("e_max_3035", "e_max_3035"), ("n_min_3035", "n_min_3035"), ("n_max_3035", "n_max_3035")):
                assert abs(t[a][q] - r[b]) < 1e-6             # libm vs numpy transcendentals: sub-micrometre
            lon, lat = GO.mercator_to_lonlat(r["geometry"][0], r["geometry"][3])
            assert abs(t["lon_min"][q] - lon) < 1e-12 and abs(t["lat_max"][q] - lat) < 1e-12
        k += c
    assert k == rows.shape[0]
    neg = (t["xmin"] < 0) | (t["ymin"] < 0)
    assert neg.any(), "the case int() truncation differs from floor was not exercised"


def test_label_text_roundtrip_equals_loadtxt(tmp_path, golden):
    stems, counts, rows32 = _random_labels(golden, 3, 11)
    o = 0
    for s, c in zip(stems, counts):
        postprocess.write_label_file(str(tmp_path), s, rows32[o:o + c], True)
        o += c
    st, cn, rows = geocode.label_dir_rows(str(tmp_path))
    assert sorted(set(stems)) == st
    k = 0
    for s, c in zip(st, cn):
        ref = np.loadtxt(os.path.join(str(tmp_path), s + ".txt"), ndmin=2)
        assert np.array_equal(ref, rows[k:k + c])
        k += c


def test_geojson_and_errors(tmp_path, golden):
    stems, counts, rows32 = _random_labels(golden, 4, 3)
    rows = rows32.astype(np.float64)
    t = geocode.geocode_detections(stems, counts, rows, golden["bboxes"])
    out = tmp_path / "d.geojson"
    assert geocode.write_geojson(str(out), stems, t) == rows.shape[0]
    d = json.load(open(out))
    ring = d["features"][0]["geometry"]["coordinates"][0]
    assert len(d["features"]) == rows.shape[0] and ring[0] == ring[-1] and len(ring) == 5
    assert ring[0][0] == t["lon_max"][0] and ring[0][1] == t["lat_min"][0]      # shapely box() ring starts at (maxx, miny)
    assert set(d["features"][0]["properties"]) >= {"image", "xmin", "xmax", "ymin", "ymax", "type", "year", "det_conf"}
    with pytest.raises(ValueError):
        geocode.geocode_detections(["no_underscores"], [1], rows[:1], golden["bboxes"])
    with pytest.raises(KeyError):
        geocode.geocode_detections(["X2020_999999_0_0"], [1], rows[:1], golden["bboxes"])
    with pytest.raises(ValueError):
        geocode.geocode_detections(stems, counts + 1, rows, golden["bboxes"])


def test_wanted_bboxes_csv_loader(tmp_path, golden):
    p = tmp_path / "wb.csv"
    with open(p, "w") as f:
        f.write(",geometry\n")
        for k, (x0, y0, x1, y1) in golden["bboxes"].items():
            f.write(f'{k},"POLYGON (({x1!r} {y0!r}, {x1!r} {y1!r}, {x0!r} {y1!r}, {x0!r} {y0!r}, {x1!r} {y0!r}))"\n')
    assert geocode.load_wanted_bboxes(str(p)) == golden["bboxes"]


def test_batch_throughput_vs_scalar_loop(golden, capsys):
    """Measurement beside parity: the batch op against the per-detection loop (the oracle's form = the reference's), same rows."""
    import time
    stems, counts, rows32 = _random_labels(golden, 400, 5)
    rows = rows32.astype(np.float64)
    t0 = time.perf_counter()
    geocode.geocode_detections(stems, counts, rows, golden["bboxes"])
    dt_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    k = 0
    for s, c in zip(stems, counts):
        GO.geocode_label_rows(s + ".txt", rows[k:k + c], golden["bboxes"])
        k += c
    dt_loop = time.perf_counter() - t0
    with capsys.disabled():
        print(f"\n[geocode] {rows.shape[0]} detections: batch {rows.shape[0] / dt_batch:,.0f}/s, scalar loop {rows.shape[0] / dt_loop:,.0f}/s "
              f"({dt_loop / dt_batch:.1f}x)")
    assert dt_batch * 2 < dt_loop


def test_tile_name_conventions_match_reference_names():
    """Real tile names from the reference's data files (tests/golden/g8_tile_names.json, cut from reference output/cf_images.csv and
    output/humanlabels.geojson): our parser reads them, our generator writes the same form, offsets are multiples of 1024 below 6144
    (reference src/utils.py:372-389; src/load_data/tile_tifs.py:13-47)."""
    from aquaculture_amd import tiles
    with open(os.path.join(HERE, "golden", "g8_tile_names.json")) as f:
        names = json.load(f)["names"]
    assert len(names) > 100
    ind, xo, yo, yr = geocode.parse_stems([n[:-5] for n in names])
    for n, i, x, y, year in zip(names, ind, xo, yo, yr):
        spec = tiles.parse_tile_name(n)
        assert (int(spec["bbox_ind"]), int(spec["x_offset"]), int(spec["y_offset"]), int(spec["year"])) == (i, x, y, year)
        assert 2000 <= year <= 2021 and x % 1024 == 0 and y % 1024 == 0 and 0 <= x < 6144 and 0 <= y < 6144
        # our synthetic-name generator produces the reference's form for the same (year, scene, offsets)
        k = int(i) * 36 + (y // 1024) * 6 + (x // 1024)
        assert tiles.tile_name(k, year=year) == n
