"""z-layer selection strategies (fusion.select_z_layers) against the known answers of the reference's own
test_z_layer_selection.py (:31-135)."""
import pytest

from image_stitcher_amd.fusion import select_z_layers, z_selection_name


def test_middle_layer():
    assert select_z_layers("middle", 5) == [2]       # middle of 0..4
    assert select_z_layers("middle", 4) == [2]       # floor division
    assert select_z_layers("middle", 1) == [0]
    for bad in (0, -1):
        with pytest.raises(ValueError):
            select_z_layers("middle", bad)


def test_all_layers():
    assert select_z_layers("all", 5) == [0, 1, 2, 3, 4]


def test_specific_layer():
    assert select_z_layers(0, 5) == [0]
    assert select_z_layers(3, 5) == [3]
    assert select_z_layers(4, 5) == [4]
    assert select_z_layers("2", 5) == [2]            # numeric strings (z_layer_selection.py:181-184)
    for bad in (-1, 5):
        with pytest.raises(ValueError):
            select_z_layers(bad, 5)


def test_unknown_strategies():
    with pytest.raises(ValueError):
        select_z_layers("invalid_strategy", 5)
    with pytest.raises(TypeError):
        select_z_layers(2.5, 5)


def test_names():
    assert z_selection_name("all") == "all_layers"
    assert z_selection_name("middle") == "middle_layer"
    assert z_selection_name(2) == "specific_layer_2"
