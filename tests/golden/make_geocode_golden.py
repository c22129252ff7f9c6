#!/usr/bin/env python3
"""Cuts tests/golden/g7_geocode.json out of the reference's own DATA files (run in the build container, where /root/reference exists):
  * /root/reference/output/humanlabels.geojson : image name, pixel box, resulting EPSG:3857 polygon, `im_center` (lat, lon)
  * /root/reference/data/wanted_bboxes.csv     : EPSG:3857 bounds of the 6144-px parent scenes, by bbox_ind
Every 8th feature is kept (518 of 4142) with the bbox rows they reference.  Data only: no reference source text is copied."""
import csv
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    bounds = {}
    with open("/root/reference/data/wanted_bboxes.csv") as f:
        r = csv.reader(f)
        next(r)
        for idx, wkt in r:
            nums = [float(v) for v in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", wkt)]
            bounds[int(idx)] = [min(nums[0::2]), min(nums[1::2]), max(nums[0::2]), max(nums[1::2])]
    d = json.load(open("/root/reference/output/humanlabels.geojson"))
    feats, used = [], {}
    for ft in d["features"][::8]:
        p = ft["properties"]
        c = ft["geometry"]["coordinates"][0]
        xs, ys = sorted(set(v[0] for v in c)), sorted(set(v[1] for v in c))
        if len(xs) != 2 or len(ys) != 2:
            continue
        ind = int(p["image"].replace(".jpeg", "").split("_")[1])
        used[ind] = bounds[ind]
        lat, lon = [float(v) for v in p["im_center"].split(",")]
        feats.append({"image": p["image"], "pix": [p["xmin"], p["ymin"], p["xmax"], p["ymax"]], "year": p["year"],
                      "poly_bounds_3857": [xs[0], ys[0], xs[1], ys[1]], "im_center_latlon": [lat, lon]})
    out = {"source": "reference output/humanlabels.geojson (every 8th feature) + data/wanted_bboxes.csv", "large_tif_size": 6144,
           "wanted_bboxes": {str(k): v for k, v in sorted(used.items())}, "features": feats}
    with open(os.path.join(HERE, "g7_geocode.json"), "w") as f:
        json.dump(out, f)
    print(len(feats), "features,", len(used), "bbox rows")


if __name__ == "__main__":
    main()
