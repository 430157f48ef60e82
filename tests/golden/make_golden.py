#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own pairwise test fixtures.

Run in the BUILD container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

For every pairwise regression case of the reference (test/test_utils.jl:77-89,101-114:
sgVerify1..17, sgNetworkVerify1..3) it stores

  * the solver-facing inputs AFTER the reference's I/O layer (src/io.jl, out of scope for this
    repo) and BEFORE graph construction: conductance cell map, polygon map, focal point
    (row, col, id) triplets, include/exclude pair table, connection flags, output flags;
  * the golden resistance matrix from test/output_verify/<case>_resistances.out
    (legacy Circuitscape outputs, 10 significant digits).

The file readers below restate only as much of src/io.jl as is needed to ingest the fixtures
(read_cellmap io.jl:91-111, read_polymap io.jl:158-194, read_point_map io.jl:196-249,
read_included_pairs io.jl:328-385, load_graph/read_focal_points io.jl:48-82, update! io.jl:511-515).
GeoTIFF inputs are not decoded (no GDAL here): sgVerify1 references polygons.tif, for which the
reference ships an identical polygons.asc next to it; that file is used instead.
"""
import configparser
import gzip
import json
import math
import os
import sys

import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))


def _open(path):
    return gzip.open(path, "rt") if path.lower().endswith("gz") else open(path, "r")


def read_aagrid(path):
    """ASCII grid -> (array float64 with nodata mapped to -9999, meta)  (io.jl:517-555 + get_raster_meta)."""
    with _open(path) as f:
        lines = f.read().splitlines()
    meta = {}
    k = 0
    while k < len(lines):
        tok = lines[k].split()
        if len(tok) >= 2 and tok[0][0].isalpha():
            meta[tok[0].lower()] = float(tok[1])
            k += 1
        else:
            break
    rows = [ln.split() for ln in lines[k:] if ln.strip()]
    arr = np.array(rows, dtype=np.float64)
    nodata = meta.get("nodata_value", -9999.0)
    arr[arr == nodata] = -9999.0
    arr[np.isnan(arr)] = -9999.0
    assert arr.shape == (int(meta["nrows"]), int(meta["ncols"])), (path, arr.shape, meta)
    return arr, meta


def resolve(path):
    p = os.path.join(REF, path)
    if p.endswith(".tif.gz"):
        p = p[:-3]
    if p.endswith(".tif"):
        alt = p[:-4] + ".asc"
        assert os.path.exists(alt), "no .asc twin for " + p
        return alt
    return p


def guess_type(path):
    with _open(path) as f:
        hdr = f.readline()
    h = hdr.lower()
    if h.startswith("ncols"):
        return "aagrid"
    if hdr.startswith("min"):
        return "pairs_aagrid"
    if hdr.startswith("mode"):
        return "pairs"
    return "txtlist"


def read_cellmap(path, is_res):
    cm, meta = read_aagrid(path)
    nod = cm == -9999
    if is_res:
        assert not np.any(cm == 0), "zero resistance"
        g = 1.0 / cm
    else:
        g = cm.copy()
    g[nod] = 0.0
    return g, meta


def read_polymap(path, nodata_as=0):
    pm, meta = read_aagrid(path)
    if nodata_as != -1:
        pm[pm == -9999] = nodata_as
    return pm, meta


def read_point_map(path, hb):
    """io.jl:196-249 -> 1-based (i, j, v) sorted by v (stable)."""
    t = guess_type(path)
    nrows = int(hb["nrows"])
    if t == "txtlist":
        with _open(path) as f:
            rows = [ln.split() for ln in f if ln.strip()]
        a = np.array(rows, dtype=np.float64)
        v = a[:, 0]
        X = a[:, 1]
        Y = a[:, 2]
        i = np.ceil(nrows - (Y - hb["yllcorner"]) / hb["cellsize"]).astype(np.int64)
        j = np.ceil((X - hb["xllcorner"]) / hb["cellsize"]).astype(np.int64)
    else:
        pm, _ = read_polymap(path)
        jj, ii = np.nonzero(pm.T)  # column-major order like findall on a Julia matrix
        i = ii + 1
        j = jj + 1
        v = pm[ii, jj]
    keep = v >= 0
    i, j, v = i[keep], j[keep], v[keep]
    order = np.argsort(v, kind="stable")
    return i[order].astype(int).tolist(), j[order].astype(int).tolist(), v[order].astype(int).tolist()


def read_included_pairs(path):
    t = guess_type(path)
    if t == "pairs_aagrid":
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.strip()]
        minval = float(lines[0].split()[1])
        maxval = float(lines[1].split()[1])
        tab = np.array([ln.split() for ln in lines[2:]], dtype=np.float64)
        point_ids = tab[1:, 0].astype(int).tolist()
        m = tab[1:, 1:].copy()
        m[m > maxval] = 0
        binm = (m >= minval).astype(int)
        return {"mode": "include", "point_ids": point_ids, "matrix": binm.tolist()}
    assert t == "pairs", path
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    mode = lines[0].split()[1]
    prs = np.array([ln.split() for ln in lines[1:]], dtype=np.float64).astype(int).reshape(-1, 2)
    ids = sorted(set(prs.ravel().tolist()) - {0})
    mat = np.zeros((len(ids), len(ids)), dtype=int)
    for a, b in prs:
        if a in ids and b in ids:
            mat[ids.index(a), ids.index(b)] = 1
            mat[ids.index(b), ids.index(a)] = 1
    return {"mode": mode, "point_ids": ids, "matrix": mat.tolist()}


def parse_ini(path):
    cp = configparser.ConfigParser()
    cp.optionxform = str
    cp.read(path)
    d = {}
    for s in cp.sections():
        for k, v in cp.items(s):
            d[k] = v.strip()
    return d


def truthy(d, k):
    return d.get(k, "false") in ("True", "true", "1")


def read_expected(name):
    p = os.path.join(REF, "output_verify", name + "_resistances.out")
    with open(p) as f:
        rows = [[float(x) for x in ln.split()] for ln in f if ln.strip()]
    return rows


def raster_case(idx):
    name = "sgVerify%d" % idx
    d = parse_ini(os.path.join(REF, "input/raster/pairwise/%d/%s.ini" % (idx, name)))
    g, hb = read_cellmap(resolve(d["habitat_file"]), truthy(d, "habitat_map_is_resistances"))
    out = {
        "name": name,
        "kind": "raster",
        "ini_solver": d.get("solver", "cg+amg"),
        "options": {
            "connect_four_neighbors_only": truthy(d, "connect_four_neighbors_only"),
            "connect_using_avg_resistances": truthy(d, "connect_using_avg_resistances"),
            "use_polygons": truthy(d, "use_polygons"),
            "use_mask": truthy(d, "use_mask"),
            "use_included_pairs": truthy(d, "use_included_pairs"),
            "write_volt_maps": truthy(d, "write_volt_maps"),
            "write_cur_maps": truthy(d, "write_cur_maps"),
            "write_cum_cur_map_only": truthy(d, "write_cum_cur_map_only"),
            "write_max_cur_maps": truthy(d, "write_max_cur_maps"),
        },
    }
    polymap = None
    if out["options"]["use_polygons"]:
        pm, _ = read_polymap(resolve(d["polygon_file"]))
        polymap = pm.astype(int).tolist()
    if out["options"]["use_mask"]:
        mk, _ = read_polymap(resolve(d["mask_file"]))
        g = g * (mk > 0)
    out["cellmap"] = g.tolist()
    out["polymap"] = polymap
    out["points_rc"] = read_point_map(resolve(d["point_file"]), hb)
    out["included_pairs"] = (
        read_included_pairs(resolve(d["included_pairs_file"])) if out["options"]["use_included_pairs"] else None
    )
    out["expected"] = read_expected(name)
    # scope row N1: golden maps (cumulative / maximum current map, per-pair current and voltage maps of up to 4 pairs)
    ov = os.path.join(REF, "output_verify")
    maps = {}
    for key, fn in (("cum_curmap", name + "_cum_curmap.asc"), ("max_curmap", name + "_max_curmap.asc")):
        if os.path.exists(os.path.join(ov, fn)):
            maps[key] = read_aagrid(os.path.join(ov, fn))[0].tolist()
    pairs = sorted(f[len(name) + 8:-4] for f in os.listdir(ov) if f.startswith(name + "_curmap_") and f.endswith(".asc"))
    maps["pairs"] = []
    for pr in pairs[:4]:
        entry = {"pair": [int(x) for x in pr.split("_")],
                 "curmap": read_aagrid(os.path.join(ov, "%s_curmap_%s.asc" % (name, pr)))[0].tolist()}
        vf = os.path.join(ov, "%s_voltmap_%s.asc" % (name, pr))
        if os.path.exists(vf):
            entry["voltmap"] = read_aagrid(vf)[0].tolist()
        maps["pairs"].append(entry)
    out["options"]["log_transform_maps"] = truthy(d, "log_transform_maps")
    out["options"]["set_null_currents_to_nodata"] = truthy(d, "set_null_currents_to_nodata")
    out["options"]["set_null_voltages_to_nodata"] = truthy(d, "set_null_voltages_to_nodata")
    out["maps"] = maps
    return out


def network_case(idx):
    name = "sgNetworkVerify%d" % idx
    d = parse_ini(os.path.join(REF, "input/network/%s.ini" % name))
    with open(resolve(d["habitat_file"])) as f:
        e = np.array([ln.split() for ln in f if ln.strip()], dtype=np.float64)
    i = e[:, 0].astype(int)
    j = e[:, 1].astype(int)
    v = e[:, 2].copy()
    mn = min(i.min(), j.min())
    assert mn <= 1
    if mn == 0:
        i = i + 1
        j = j + 1
    if truthy(d, "habitat_map_is_resistances"):
        v = 1.0 / v
    with open(resolve(d["point_file"])) as f:
        fp = np.array([float(x) for x in f.read().split()]).astype(int)
    if fp.min() == 0:
        fp = fp + 1
    ov = os.path.join(REF, "output_verify")

    def table(fn):
        with open(os.path.join(ov, fn)) as f:
            return [[float(x) for x in ln.split()] for ln in f if ln.strip()]

    pairs = sorted(f[len(name) + 17:-4] for f in os.listdir(ov)
                   if f.startswith(name + "_branch_currents_") and f.endswith(".txt") and not f.endswith("_cum.txt"))
    tables = {"pairs": []}
    for pr in pairs[:4]:
        tables["pairs"].append({"pair": [int(x) for x in pr.split("_")],           # 0-based ids, as in the golden files
                                "branch": table("%s_branch_currents_%s.txt" % (name, pr)),
                                "node": table("%s_node_currents_%s.txt" % (name, pr)),
                                "voltages": table("%s_voltages_%s.txt" % (name, pr))})
    tables["branch_cum"] = table(name + "_branch_currents_cum.txt")
    tables["node_cum"] = table(name + "_node_currents_cum.txt")
    return {
        "tables": tables,
        "name": name,
        "kind": "network",
        "ini_solver": d.get("solver", "cg+amg"),
        "edges_i": i.tolist(),
        "edges_j": j.tolist(),
        "edges_v": v.tolist(),
        "focal": fp.tolist(),
        "expected": read_expected(name),
    }


def network_advanced_case(idx):
    """mgNetworkVerify<idx>: network advanced mode (test/test_utils.jl:91-99) -- inputs after get_network_data
    (io.jl:387-418: load_graph + read_point_strengths) and the golden node voltages."""
    name = "mgNetworkVerify%d" % idx
    d = parse_ini(os.path.join(REF, "input/network/%s.ini" % name))
    with open(resolve(d["habitat_file"])) as f:
        e = np.array([ln.split() for ln in f if ln.strip()], dtype=np.float64)
    i = e[:, 0].astype(int)
    j = e[:, 1].astype(int)
    v = e[:, 2].copy()
    zero_based = min(i.min(), j.min()) == 0
    if zero_based:
        i = i + 1
        j = j + 1
    if truthy(d, "habitat_map_is_resistances"):
        v = 1.0 / v

    def strengths(path):  # read_point_strengths, io.jl:84-89
        with open(path) as f:
            a = np.array([ln.split() for ln in f if ln.strip()], dtype=np.float64)
        if a[:, 0].min() == 0 or zero_based:
            a[:, 0] += 1
        return a.tolist()

    with open(os.path.join(REF, "output_verify", name + "_voltages.txt")) as f:
        exp = [[float(x) for x in ln.split()] for ln in f if ln.strip()]
    return {
        "name": name, "kind": "network_advanced", "ini_solver": d.get("solver", "cg+amg"),
        "edges_i": i.tolist(), "edges_j": j.tolist(), "edges_v": v.tolist(),
        "sources": strengths(resolve(d["source_file"])), "grounds": strengths(resolve(d["ground_file"])),
        "ground_file_is_resistances": truthy(d, "ground_file_is_resistances"),
        "remove_src_or_gnd": d.get("remove_src_or_gnd", "keepall"),
        "expected_voltages": exp,  # [node (0-based in the golden file), voltage]
    }


def _txt_list(path, hb):
    """_txt_list_reader (io.jl:315-326): rows (value, X, Y) -> (value, row, col), 1-based."""
    with _open(path) as f:
        a = np.array([ln.split() for ln in f if ln.strip()], dtype=np.float64)
    nrows = int(hb["nrows"])
    r = np.ceil(nrows - (a[:, 2] - hb["yllcorner"]) / hb["cellsize"]).astype(int)
    c = np.ceil((a[:, 1] - hb["xllcorner"]) / hb["cellsize"]).astype(int)
    return a[:, 0], r, c


def read_source_and_ground_maps(d, hb):
    """io.jl:252-313: source map (nodata -> 0), ground map (nodata -> 0, resistances inverted so that a zero
    resistance becomes an infinite conductance = direct ground), use_unit_currents / use_direct_grounds."""
    shape = (int(hb["nrows"]), int(hb["ncols"]))
    gpath = resolve(d["ground_file"])
    if guess_type(gpath) == "aagrid":
        ground, _ = read_polymap(gpath, nodata_as=-1)
    else:
        v, r, c = _txt_list(gpath, hb)
        ground = -9999.0 * np.ones(shape)
        ground[r - 1, c - 1] = v
    spath = resolve(d["source_file"])
    if guess_type(spath) == "aagrid":
        source, _ = read_polymap(spath)
        source[source == -9999] = 0
    else:
        v, r, c = _txt_list(spath, hb)
        source = np.zeros(shape)
        source[r - 1, c - 1] = v
    nod = ground == -9999
    if truthy(d, "ground_file_is_resistances"):
        with np.errstate(divide="ignore"):
            ground = 1.0 / ground
    ground[nod] = 0
    if truthy(d, "use_unit_currents"):
        source[source != 0] = 1
    if truthy(d, "use_direct_grounds"):
        ground[ground != 0] = np.inf
    return source, ground


def _jsonable(a):
    """nested lists with +-inf spelled as strings (strict JSON has no Infinity)."""
    a = np.asarray(a, dtype=np.float64)
    return [[("inf" if v == np.inf else "-inf" if v == -np.inf else float(v)) for v in row] for row in a]


def raster_advanced_case(idx):
    """mgVerify<idx>: raster advanced mode (test/test_utils.jl:116-121) -- inputs after load_raster_data
    (io.jl:420-510) and the golden voltage / current maps."""
    name = "mgVerify%d" % idx
    d = parse_ini(os.path.join(REF, "input/raster/advanced/%d/%s.ini" % (idx, name)))
    g, hb = read_cellmap(resolve(d["habitat_file"]), truthy(d, "habitat_map_is_resistances"))
    opts = {k: truthy(d, k) for k in (
        "connect_four_neighbors_only", "connect_using_avg_resistances", "use_polygons", "use_mask", "write_volt_maps",
        "write_cur_maps", "write_cum_cur_map_only", "write_max_cur_maps", "log_transform_maps",
        "set_null_currents_to_nodata", "set_null_voltages_to_nodata")}
    opts["remove_src_or_gnd"] = d.get("remove_src_or_gnd", "keepall")
    polymap = None
    if opts["use_polygons"]:
        polymap = read_polymap(resolve(d["polygon_file"]))[0].astype(int).tolist()
    if opts["use_mask"]:
        mk, _ = read_polymap(resolve(d["mask_file"]))
        g = g * (mk > 0)
    source, ground = read_source_and_ground_maps(d, hb)
    ov = os.path.join(REF, "output_verify")
    exp = {}
    for key in ("voltmap", "curmap"):
        fn = os.path.join(ov, "%s_%s.asc" % (name, key))
        if os.path.exists(fn):
            exp[key] = read_aagrid(fn)[0].tolist()
    return {"name": name, "kind": "raster_advanced", "ini_solver": d.get("solver", "cg+amg"), "options": opts,
            "cellmap": g.tolist(), "polymap": polymap, "source_map": _jsonable(source), "ground_map": _jsonable(ground),
            "expected": exp}


def onetoall_case(kind, idx):
    """oneToAllVerify<idx> / allToOneVerify<idx> (test/test_utils.jl:123-140): inputs after load_raster_data and the
    golden per-point resistances, per-point voltage / current maps, cumulative and maximum current maps."""
    name = ("oneToAllVerify%d" if kind == "one_to_all" else "allToOneVerify%d") % idx
    d = parse_ini(os.path.join(REF, "input/raster/%s/%d/%s.ini" % (kind, idx, name)))
    assert d["scenario"] == ("one-to-all" if kind == "one_to_all" else "all-to-one")
    g, hb = read_cellmap(resolve(d["habitat_file"]), truthy(d, "habitat_map_is_resistances"))
    opts = {k: truthy(d, k) for k in (
        "connect_four_neighbors_only", "connect_using_avg_resistances", "use_polygons", "use_mask", "write_volt_maps",
        "write_cur_maps", "write_cum_cur_map_only", "write_max_cur_maps", "log_transform_maps",
        "set_null_currents_to_nodata", "set_null_voltages_to_nodata", "use_included_pairs",
        "use_variable_source_strengths")}
    polymap = None
    if opts["use_polygons"]:
        polymap = read_polymap(resolve(d["polygon_file"]))[0].astype(int).tolist()
    if opts["use_mask"]:
        mk, _ = read_polymap(resolve(d["mask_file"]))
        g = g * (mk > 0)
    strengths = None
    if opts["use_variable_source_strengths"]:
        with open(resolve(d["variable_source_file"])) as f:
            a = np.array([ln.split() for ln in f if ln.strip()], dtype=np.float64)
        if a[:, 0].min() == 0:   # read_point_strengths, io.jl:84-89
            a[:, 0] += 1
        strengths = a.tolist()
    ov = os.path.join(REF, "output_verify")
    with open(os.path.join(ov, name + "_resistances.out")) as f:
        res = [[float(x) for x in ln.split()] for ln in f if ln.strip()]
    maps = {"points": {}}
    for key in ("cum_curmap", "max_curmap"):
        fn = os.path.join(ov, "%s_%s.asc" % (name, key))
        if os.path.exists(fn):
            maps[key] = read_aagrid(fn)[0].tolist()
    for fn in sorted(os.listdir(ov)):
        for key in ("curmap", "voltmap"):
            pre = "%s_%s_" % (name, key)
            if fn.startswith(pre) and fn.endswith(".asc"):
                maps["points"].setdefault(fn[len(pre):-4], {})[key] = read_aagrid(os.path.join(ov, fn))[0].tolist()
    return {"name": name, "kind": kind, "ini_solver": d.get("solver", "cg+amg"), "options": opts,
            "cellmap": g.tolist(), "polymap": polymap, "points_rc": read_point_map(resolve(d["point_file"]), hb),
            "included_pairs": (read_included_pairs(resolve(d["included_pairs_file"]))
                               if opts["use_included_pairs"] else None),
            "strengths": strengths, "expected": res, "maps": maps}


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    cases = ([raster_case(k) for k in range(1, 18)] + [network_case(k) for k in range(1, 4)] +
             [network_advanced_case(k) for k in range(1, 4)] + [raster_advanced_case(k) for k in range(1, 7)] +
             [onetoall_case("one_to_all", k) for k in range(1, 14)] +
             [onetoall_case("all_to_one", k) for k in range(1, 13)])
    for c in cases:
        with open(os.path.join(OUT, c["name"] + ".json"), "w") as f:
            json.dump(c, f, separators=(",", ":"))
        print("wrote", c["name"])
    # mgVerify7 (355 x 481 cells, 25 sources, 5574 finite grounds): the reference ships its inputs and its golden current
    # map but its suite stops at mgVerify6 (test/test_utils.jl:117); the only golden of the reference at a landscape's size.
    # Stored gzipped (mtime 0: the bytes do not depend on when the script ran).
    c = raster_advanced_case(7)
    with open(os.path.join(OUT, c["name"] + ".json.gz"), "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(json.dumps(c, separators=(",", ":")).encode())
    print("wrote", c["name"], "(gzipped)")


if __name__ == "__main__":
    main()
