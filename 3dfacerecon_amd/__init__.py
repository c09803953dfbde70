"""MI355X-native render_depth + 3DMM-decode hot path of Cogito2012/3DFaceRecon.

The directory name starts with a digit, so import it with importlib::

    import importlib
    ops = importlib.import_module("3dfacerecon_amd.rendering_layer.ops")

or put this directory on sys.path and use the reference's own module paths
(``from rendering_layer.ops import render_depth``, ``from nets.network import FaceRecNet``).
"""
