"""The HIP path against every call the reference's own tests make to the hot-path functions, through the
package API that mirrors the reference's (and, below it, the C ABI).  Fixture: tests/golden/reference_test_calls.npz,
recorded from /root/reference/tests/{bars,features} by oracle/record_reference_tests.py; each failure message names the
reference test the call came from."""
import pytest

from tests import _refcalls as R

pytestmark = pytest.mark.gpu

SKIP = {
    "calc_volume_percentage_above_poc": "not exported as a stand-alone function: the share above the POC is evaluated "
                                        "inside k_volume_profile with the POC it finds itself; the recorded calls pass "
                                        "an arbitrary POC.  The 4 volume_profile_rolling records cover the kernel.",
}


def _realized_volatility(attrs, frame, kwargs):
    from finmlkit_amd.feature.transforms import RealizedVolatility
    rv = RealizedVolatility(window=attrs["window"], is_sample=attrs["is_sample"], input_col=attrs["requires"][0])
    assert list(rv.produces) == attrs["produces"]
    return rv(frame)


def _table():
    from finmlkit_amd.bar import base, kit, logic, utils
    from finmlkit_amd.bar.data_model import TradesData
    from finmlkit_amd.feature.core import utils as futils
    from finmlkit_amd.feature.core import volatility, volume
    table = {
        "_time_bar_indexer": logic._time_bar_indexer,
        "_tick_bar_indexer": logic._tick_bar_indexer,
        "_volume_bar_indexer": logic._volume_bar_indexer,
        "_dollar_bar_indexer": logic._dollar_bar_indexer,
        "_cusum_bar_indexer": logic._cusum_bar_indexer,
        "TradesData": TradesData,
        "TimeBarKit._comp_bar_close": kit.TimeBarKit, "TickBarKit._comp_bar_close": kit.TickBarKit,
        "VolumeBarKit._comp_bar_close": kit.VolumeBarKit, "DollarBarKit._comp_bar_close": kit.DollarBarKit,
        "CUSUMBarKit._comp_bar_close": kit.CUSUMBarKit,
        "comp_bar_ohlcv": base.comp_bar_ohlcv,
        "comp_bar_directional_features": base.comp_bar_directional_features,
        "comp_bar_footprints": base.comp_bar_footprints,
        "comp_footprint_features": base.comp_footprint_features,
        "comp_bar_trade_size_features": base.comp_bar_trade_size_features,
        "comp_price_tick_size": utils.comp_price_tick_size,
        "comp_trade_side_vector": utils.comp_trade_side_vector,
        "merge_split_trades": utils.merge_split_trades,
        "footprint_to_dataframe": utils.footprint_to_dataframe,
        "comp_lagged_returns": futils.comp_lagged_returns,
        "ewms": volatility.ewms,
        "realized_vol": volatility.realized_vol,
        "volume_profile_rolling": volume.volume_profile_rolling,
        # the reference's tests call the two backends of the transform directly; both are replayed through the one
        # (HIP) backend here, the pandas recordings at the reference's own pd-vs-compiled tolerance (rtol 1e-10)
        "RealizedVolatility._pd": _realized_volatility,
        "RealizedVolatility._nb": _realized_volatility,
        "ewmst": volatility.ewmst, "ewmst_mean0": volatility.ewmst_mean0,
    }
    return table


def test_hip_path_replays_reference_test_calls():
    done, skipped = R.replay(_table(), SKIP)
    assert done == 156 and skipped == {"calc_volume_percentage_above_poc": 4}, (done, skipped)    # of 160 recorded calls


def test_hip_path_replays_edge_sweep():
    """Degenerate inputs of our own through the reference's functions (oracle/edge_sweep.py -> edge_calls.npz), replayed
    through the package: results under the contract of DESIGN.md 5, exceptions by type."""
    done, skipped = R.replay(_table(), SKIP, path=R.EDGE_PATH, match_message=False)
    assert done == 175 and skipped == {"not comparable": 15}, (done, skipped)    # 137 function calls + 39 TradesData(...)
