"""Modality / band tables for MP-MAE pretraining (configuration data).

Same names and meaning as the reference's tables (/root/reference/MODALITIES.py:56-189):
INP_MODALITIES, OUT_MODALITIES (order = loss / log_var order), MODALITIES_FULL,
MODALITY_TASK, PIXEL_WISE_MODALITIES, plus the README's named subsets
(/root/reference/README.md:54-59).
"""
from collections import OrderedDict

_S2_ALL = "B1 B2 B3 B4 B5 B6 B7 B8A B8 B9 B10 B11 B12".split()
_S2_12 = [b for b in _S2_ALL if b != "B10"]  # L2A has no B10
_S1 = [f"{orb}_{pol}" for orb in ("asc", "desc") for pol in ("VV", "VH", "HH", "HV")]
_ERA5 = [f"{w}_{s}" for w in ("prev_month", "curr_month", "year")
         for s in ("avg_temp", "min_temp", "max_temp", "total_precip")]

MODALITIES_FULL = OrderedDict(
    sentinel2=_S2_ALL,
    sentinel2_cloudmask=["QA60"],
    sentinel2_cloudprod=["MSK_CLDPRB"],
    sentinel2_scl=["SCL"],
    sentinel1=_S1,
    aster=["elevation", "slope"],
    era5=_ERA5,
    dynamic_world=["landcover"],
    canopy_height_eth=["height", "std"],
    lat=["sin", "cos"],
    lon=["sin", "cos"],
    biome=["biome"],
    eco_region=["eco_region"],
    month=["sin_month", "cos_month"],
    esa_worldcover=["map"],
)

INP_MODALITIES = OrderedDict(sentinel2=list(_S2_12))

# order matters: it is the order of loss_dict, log_vars and the weighted-loss vector
OUT_MODALITIES = OrderedDict(
    sentinel2=list(_S2_12),
    sentinel1="all",
    aster="all",
    era5="all",
    dynamic_world="all",
    canopy_height_eth="all",
    lat="all",
    lon="all",
    biome="all",
    eco_region="all",
    month="all",
    esa_worldcover="all",
)

RGB_MODALITIES = OrderedDict(sentinel2=["B2", "B3", "B4"])

PIXEL_WISE_MODALITIES = [
    "sentinel2", "sentinel1", "aster", "canopy_height_eth", "esa_worldcover", "dynamic_world",
]
IMAGE_WISE_MODALITIES = ["biome", "eco_region", "lat", "lon", "month", "era5"]

MODALITY_TASK = dict(
    sentinel2="regression_map", sentinel1="regression_map", aster="regression_map",
    canopy_height_eth="regression_map",
    lat="regression", lon="regression", month="regression", era5="regression",
    esa_worldcover="segmentation", dynamic_world="segmentation",
    biome="classification", eco_region="classification",
)

# class counts of the categorical modalities (/root/reference/models/fcmae.py:78-91)
NUM_CLASSES = dict(biome=14, eco_region=846, esa_worldcover=11, dynamic_world=9)


def subset(name: str) -> OrderedDict:
    """Named OUT_MODALITIES subsets of the reference README: all_mod, pix_mod, img_mod, S2."""
    if name == "all_mod":
        return OrderedDict(OUT_MODALITIES)
    if name == "pix_mod":
        return OrderedDict((k, v) for k, v in OUT_MODALITIES.items() if k in PIXEL_WISE_MODALITIES)
    if name == "img_mod":
        return OrderedDict((k, v) for k, v in OUT_MODALITIES.items()
                           if k in IMAGE_WISE_MODALITIES or k == "sentinel2")
    if name in ("S2", "s2"):
        return OrderedDict(sentinel2=list(_S2_12))
    raise KeyError(name)
