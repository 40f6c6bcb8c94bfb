"""Replay of the calls the REFERENCE'S OWN tests make to the hot-path functions (tests/golden/reference_test_calls.npz,
recorded by oracle/record_reference_tests.py from /root/reference/tests/{bars,features}).

Every record cites the reference test that made the call and holds its arguments and the reference's result (or the
exception it raised).  `replay(target)` runs each record through a table of callables -- the oracle's (CPU tests) or the
package's (GPU tests) -- and compares under the numerical contract of DESIGN.md 5.  Records of functions a target has no
counterpart for must be listed in that target's `skip` table with a reason: nothing is dropped silently."""
import json
import os

import numpy as np

from tests import _golden as G

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_test_calls.npz")
# degenerate inputs of our own through the reference's functions (oracle/edge_sweep.py), same format
EDGE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edge_calls.npz")


def load(path=PATH):
    d = np.load(path, allow_pickle=False)
    return json.loads(bytes(d["__manifest__"]).decode()), d


def dec(e, d):
    t = e["t"]
    if t == "py":
        return e["v"]
    if t == "int":
        return int(e["v"])
    if t == "float":
        return float(e["v"])
    if t == "nd":
        return np.array(d[e["k"]])
    if t == "list":
        v = [dec(x, d) for x in e["v"]]
        if e["kind"].startswith("pandas:"):            # an Index argument: datetimes were stored as int64 ns
            import pandas as pd
            return pd.DatetimeIndex(np.asarray(v[0], dtype="datetime64[ns]")) if e["kind"] == "pandas:DatetimeIndex" \
                else pd.Index(v[0])
        return tuple(v) if e["kind"] == "tuple" else v
    if t == "ragged":
        vals, off = dec(e["values"], d), dec(e["offsets"], d)
        return [vals[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    if t == "dict":
        return {k: dec(x, d) for k, x in e["v"].items()}
    if t == "timedelta_ns":
        import pandas as pd
        return pd.Timedelta(int(e["v"]), unit="ns")
    if t == "series":
        return {"__series__": True, "name": e["name"], "index": dec(e["index"], d), "index_kind": e["index_kind"],
                "values": dec(e["values"], d)}
    if t == "df":
        return {"__df__": True, "columns": e["columns"], "index": dec(e["index"], d), "index_kind": e["index_kind"],
                "index_names": e["index_names"], "index_level_kinds": e.get("index_level_kinds"),
                "cols": [dec(c, d) for c in e["cols"]]}
    raise ValueError(f"cannot decode {t}")


def to_pandas(v):
    import pandas as pd
    ix = v["index"]
    names = v.get("index_names") or [None]
    if v["index_kind"] == "DatetimeIndex":
        ix = pd.DatetimeIndex(np.asarray(ix, dtype="datetime64[ns]"), name=names[0])
    elif v["index_kind"] != "MultiIndex" and not v.get("__series__"):
        ix = pd.Index(ix, name=names[0])
    elif v["index_kind"] == "MultiIndex":      # datetime levels are stored as int64 ns
        ix = pd.MultiIndex.from_arrays([pd.DatetimeIndex(np.asarray(a, dtype="datetime64[ns]")) if k == "DatetimeIndex" else a
                                        for a, k in zip(ix, v["index_level_kinds"])], names=v.get("index_names"))
    if v.get("__series__"):
        return pd.Series(v["values"], index=ix, name=v["name"])
    return pd.DataFrame({c: a for c, a in zip(v["columns"], v["cols"])}, index=ix)


# ---- comparison ---------------------------------------------------------------------------------
# float policy per function: "exact" | ("rtol", r) | ("atol", a); per result position / column where they differ
POLICY = {
    "_time_bar_indexer": "exact",
    "comp_bar_ohlcv": {5: ("rtol", 1e-9), None: "exact"},        # vwap: float64 sums in tree order
    "comp_bar_directional_features": "exact",
    # vp_skew = sum((p - vwap) * v) / sum(v) is identically 0 in exact arithmetic: the reference's value is rounding noise
    # of a BLAS dot product -> the absolute tolerance of tests/test_gpu_features.py / test_oracle_golden.py
    "comp_bar_footprints": {11: ("atol", 1e-6), None: "exact"},
    "comp_footprint_features": {4: ("atol", 1e-6), None: "exact"},
    "comp_bar_trade_size_features": "exact",
    "comp_price_tick_size": "exact",
    "comp_trade_side_vector": "exact",
    "merge_split_trades": "exact",
    "footprint_to_dataframe": "exact",
    "comp_lagged_returns": ("rtol", 1e-12),
    "ewms": ("rtol", 1e-9),
    "ewmst": ("rtol", 1e-9), "ewmst_mean0": ("rtol", 1e-9),
    "realized_vol": ("rtol", 1e-9),
    "volume_profile_rolling": "exact",
    "resample_bars": "exact",                  # pandas' Kahan sums row by row: nothing is reassociated
    # float64 accumulator / quotient of the Numba-typed function vs the recorded pure-Python float32 one (same split as
    # VolumePro.compute's fourth output below): within 2 ulp(float32)
    "calc_volume_percentage_above_poc": ("rtol", 2.4e-7),
    "_tick_bar_indexer": "exact", "_volume_bar_indexer": "exact", "_dollar_bar_indexer": "exact",
    "_cusum_bar_indexer": "exact",
    "TradesData": "exact",
    # API level (oracle/edge_sweep.py api_records): DataFrames / FootprintData attributes of the kits, transforms, VolumePro
    "build_ohlcv": {"vwap": ("rtol", 1e-9), None: "exact"},
    "build_directional_features": "exact",
    "build_trade_size_features": "exact",      # float32 amounts after the merge: NumPy's float32 trees, reproduced exactly
    "build_footprints": {"vp_skew": ("atol", 1e-6), None: "exact"},
    "api_transform": ("rtol", 1e-9),
    # pct_above_poc: `volume_above_poc = 0.0; += volumes[i]` (volume.py:381-384) stays float32 in the recorded pure-Python
    # mode (NEP 50) but is float64 under Numba's type unification, which this path follows -> 1 ulp(float32) apart on
    # footprints whose float32 level sums are not exact; POC / HVA / LVA must be identical
    "VolumePro.compute": {3: ("rtol", 2.4e-7), None: "exact"},
    "TimeBarKit._comp_bar_close": "exact", "TickBarKit._comp_bar_close": "exact", "VolumeBarKit._comp_bar_close": "exact",
    "DollarBarKit._comp_bar_close": "exact", "CUSUMBarKit._comp_bar_close": "exact",
    # the reference's own agreement between its pandas and its compiled backend (test_realized_volatility.py:25)
    "RealizedVolatility._pd": ("rtol", 1e-10),
    "RealizedVolatility._nb": ("rtol", 1e-9),
}


def _cmp_array(got, want, pol, what):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if want.dtype.kind in "iub" or pol == "exact":
        if want.dtype.kind in "iub":
            assert got.dtype.kind == want.dtype.kind or got.dtype.kind in "iub", f"{what}: dtype {got.dtype} vs {want.dtype}"
        np.testing.assert_array_equal(got, want, err_msg=what)
        return
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern"
    if pol[0] == "atol":
        np.testing.assert_allclose(got, want, rtol=0, atol=pol[1], equal_nan=True, err_msg=what)
    else:
        np.testing.assert_allclose(got, want, rtol=pol[1], atol=1e-300, equal_nan=True, err_msg=what)


def compare(fn, got, want, what):
    pol = POLICY[fn]
    named = isinstance(pol, dict) and any(isinstance(k, str) for k in pol)      # per column / attribute name

    def at(i):
        return pol.get(i, pol[None]) if isinstance(pol, dict) else pol

    def rec(g, w, p, path):
        if isinstance(w, dict) and (w.get("__series__") or w.get("__df__")):
            import pandas as pd
            wp = to_pandas(w)
            if w.get("__series__"):
                assert isinstance(g, pd.Series), f"{path}: {type(g)}"
                assert g.name == wp.name, f"{path}: name {g.name!r} vs {wp.name!r}"
                assert g.index.equals(wp.index), f"{path}: index"
                assert g.dtype == wp.dtype, f"{path}: dtype {g.dtype} vs {wp.dtype}"
                _cmp_array(g.to_numpy(), wp.to_numpy(), p, path)
            else:
                assert isinstance(g, pd.DataFrame), f"{path}: {type(g)}"
                assert list(map(str, g.columns)) == list(wp.columns), f"{path}: columns {list(g.columns)}"
                assert g.index.nlevels == wp.index.nlevels, f"{path}: index levels"
                assert list(g.index.names) == list(wp.index.names), f"{path}: index names {list(g.index.names)}"
                for i in range(wp.index.nlevels):
                    assert g.index.get_level_values(i).dtype == wp.index.get_level_values(i).dtype, f"{path}: index {i} dtype"
                    np.testing.assert_array_equal(np.asarray(g.index.get_level_values(i)),
                                                  np.asarray(wp.index.get_level_values(i)), err_msg=f"{path}: index {i}")
                for c in wp.columns:
                    assert g[c].dtype == wp[c].dtype, f"{path}.{c}: dtype {g[c].dtype} vs {wp[c].dtype}"
                    _cmp_array(g[c].to_numpy(), wp[c].to_numpy(), pol.get(c, pol[None]) if named else p, f"{path}.{c}")
            return
        if isinstance(w, dict):
            assert sorted(g) == sorted(w), f"{path}: keys {sorted(g)} vs {sorted(w)}"
            for k in w:
                rec(g[k], w[k], pol.get(k, pol[None]) if named else p, f"{path}.{k}")
            return
        if isinstance(w, (tuple, list)):
            assert len(g) == len(w), f"{path}: length {len(g)} vs {len(w)}"
            for i, (gi, wi) in enumerate(zip(g, w)):
                rec(gi, wi, p, f"{path}[{i}]")
            return
        if isinstance(w, np.ndarray):
            _cmp_array(g, w, p, path)
            return
        if isinstance(w, float):
            _cmp_array(np.float64(g), np.float64(w), p, path)
            return
        assert g == w, f"{path}: {g!r} vs {w!r}"

    if named:
        rec(got, want, pol[None], what)
    elif isinstance(want, tuple) and isinstance(pol, dict):
        assert len(got) == len(want), f"{what}: length {len(got)} vs {len(want)}"
        for i, (g, w) in enumerate(zip(got, want)):
            rec(g, w, at(i), f"{what}[{i}]")
    else:
        rec(got, want, at(None), what)


def replay(table, skip, path=PATH, match_message=True):
    """-> (n_replayed, n_skipped_by_fn).  Asserts on the first mismatch, naming the citing reference test.
    Records carrying a `skip_reason` (edge sweep: behaviour that exists only in the reference's pure-Python mode, or is
    garbage) are counted under "not comparable"."""
    man, d = load(path)
    # the recorded answers come from tests that passed under the reference itself -- except tests that need PyTables
    # (absent from the image): those stop with pandas' ImportError in their HDF5 part; the calls they made before are kept
    assert all("ImportError" in why for why in man["tests_not_passed"].values()), man["tests_not_passed"]
    built = {}                                          # TradesData objects of the target, by record index
    done, skipped = 0, {}
    for i, c in enumerate(man["calls"]):
        fn = c["fn"]
        if "skip_reason" in c:
            skipped["not comparable"] = skipped.get("not comparable", 0) + 1
            continue
        if fn in skip:
            skipped[fn] = skipped.get(fn, 0) + 1
            continue
        kind_key = {"kit_build": "api:kit_build", "api_transform": "api:transform", "api_volumepro": "api:volumepro"}.get(c.get("kind"))
        if kind_key:
            if kind_key in skip:
                skipped[kind_key] = skipped.get(kind_key, 0) + 1
                continue
            fn = kind_key
        assert fn in table, f"recorded function {fn} has neither a replay nor a documented skip"
        args = [dec(a, d) for a in c["args"]]
        kwargs = {k: dec(v, d) for k, v in c["kwargs"].items()}
        what = f"call {i} {fn} <- {c['test']}"
        call = table[fn]
        if c.get("kind") == "transform":
            attrs = {k: dec(v, d) for k, v in c["attrs"].items()}
            run = lambda: call(attrs, to_pandas(args[0]), kwargs)          # noqa: E731
        elif c.get("kind") == "tradesdata":
            def run(i=i, args=args, kwargs=kwargs, call=call):
                obj = call(*args, **kwargs)
                built[i] = obj
                return {"data": obj.data, "orig_timestamp_unit": obj.orig_timestamp_unit}
        elif c.get("kind") in ("kit_build", "api_transform", "api_volumepro"):
            def run(c=c, call=call):
                return call(c, lambda e: dec(e, d))
        elif c.get("kind") == "kit":
            def run(c=c, call=call):
                ctor = c["ctor"]
                assert ctor["trades"] in built, f"kit record refers to TradesData record {ctor['trades']} not replayed"
                kit = call(built[ctor["trades"]], *[dec(a, d) for a in ctor["args"]],
                           **{k: dec(v, d) for k, v in ctor["kwargs"].items()})
                return kit._comp_bar_close()
        else:
            run = lambda: call(*args, **kwargs)                            # noqa: E731
        if "raises" in c:
            import builtins
            exc = getattr(builtins, c["raises"]["type"], None) or getattr(builtins, c["raises"]["base"])
            try:
                run()
            except exc as e:
                if match_message:
                    assert str(e) == c["raises"]["msg"], f"{what}: message {str(e)!r} vs {c['raises']['msg']!r}"
            else:
                raise AssertionError(f"{what}: expected {c['raises']['type']}({c['raises']['msg']!r})")
        else:
            pfn = {"kit_build": c.get("method"), "api_transform": "api_transform", "api_volumepro": "VolumePro.compute"}.get(
                c.get("kind"), fn)
            compare(pfn, run(), dec(c["result"], d), what)
        done += 1
    return done, skipped
