"""GetTimeAndWeather / Normaliztion_1D / WorkdayOrWeekend of the Simulation shell against
values captured from the unmodified reference (simulator.py:79-95, 697-706, 833-866).  CPU only: the shell's
constructor needs no GPU."""
import numpy as np
import pandas as pd
import pytest

from helpers import load_golden
from vehicles_dispatch_simulator_amd.config.setting import *  # noqa: F401,F403  (same names as the reference's setting.py)
from vehicles_dispatch_simulator_amd.simulation import Simulation


def shell():
    return Simulation(ClusterMode="KmeansClustering", DemandPredictionMode="None", DispatchMode="Simulation", VehiclesNumber=10,
                      TimePeriods=TIMESTEP, LocalRegionBound=(104.011, 104.125, 30.618, 30.703), SideLengthMeter=800,
                      VehiclesServiceMeter=800, NeighborCanServer=False, FocusOnLocalRegion=False, Quiet=True)


def test_normalised_weather_tables_equal_reference():
    g = load_golden("tiny_kmeans")
    S = shell()
    got = np.concatenate([S.WeatherType, S.MinimumTemperature, S.MaximumTemperature, S.WindDirection, S.WindPower])
    assert got.dtype == np.float64 and np.array_equal(got, g["weather_tables"])      # same divisions: bit-identical


def test_features_over_the_month_equal_reference():
    g = load_golden("tiny_kmeans")
    S = shell()
    times = [pd.Timestamp(2016, 11, 1) + pd.Timedelta(minutes=int(m)) for m in g["tw_grid_minutes"]]
    np.testing.assert_array_equal(S.TimeAndWeatherFeatures(times), g["tw_grid"])

    class O:
        pass
    for ts, exp in zip(times[::7], g["tw_grid"][::7]):
        o = O(); o.ReleasTime = ts
        assert [float(x) for x in S.GetTimeAndWeather(o)] == exp.tolist()
    o = O(); o.ReleasTime = pd.Timestamp(2016, 12, 1)
    with pytest.raises(Exception, match="Month format error"):
        S.GetTimeAndWeather(o)
    with pytest.raises(Exception, match="Month format error"):
        S.TimeAndWeatherFeatures([o.ReleasTime])


@pytest.mark.parametrize("name", ["tiny_kmeans", "tiny_two_orders"])
def test_order_features_equal_reference(name):
    g = load_golden(name)
    S = shell()
    # orders' release minutes are relative to the first order; the fixture's first feature row gives its clock time
    first = g["o_time_weather"][0]
    t0 = pd.Timestamp(2016, 11, int(first[0]), int(first[3]), int(first[4]))
    times = [t0 + pd.Timedelta(minutes=int(m)) for m in g["o_release_min"]]
    np.testing.assert_array_equal(S.TimeAndWeatherFeatures(times), g["o_time_weather"])


def test_workday_or_weekend():
    S = shell()
    assert [S.WorkdayOrWeekend(d) for d in range(7)] == ["Workday"] * 5 + ["Weekend"] * 2
    for bad in (-1, 7, 1.0, "1"):
        with pytest.raises(Exception, match="input format error"):
            S.WorkdayOrWeekend(bad)
