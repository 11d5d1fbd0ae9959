"""Binary layouts of the GraviT types that cross the adapter boundary.

All layouts are the reference's (file:line relative to the GraviT tree):
  Ray       80 B, align 16   src/gvt/render/actor/Ray.h:68-96
  Material  92 B             src/gvt/render/data/primitives/Material.h:59-90
  Light     64 B tagged POD  (CPU reference uses virtual classes, scene/Light.h:46-104; POD precedent
                              adapter/optix/Light.cuh:44-112)
"""
import numpy as np

RAY_PRIMARY, RAY_SHADOW, RAY_SECONDARY = 0, 1, 2  # Ray.h:50-55
LIGHT_POINT, LIGHT_AREA, LIGHT_AMBIENT = 0, 1, 2
LAMBERT, PHONG, BLINN, EMBREE_METAL, EMBREE_VELVET, EMBREE_MATTE = 0, 1, 2, 3, 4, 5  # Material.h:49-57
NORMALS_FLAT, NORMALS_SMOOTH = 0, 1  # EmbreeMeshAdapter.cpp:75 (FLAT_SHADING) vs goldens / EmbreeStream / OptiX
RAY_EPSILON = np.float32(1.0e-6)  # Ray.cpp:33
FLT_MAX = np.float32(np.finfo(np.float32).max)

RAY_DTYPE = np.dtype(
    {
        "names": ["origin", "t_min", "direction", "t_max", "color", "t", "id", "depth", "w", "type", "rng", "known"],
        "formats": [("<f4", 3), "<f4", ("<f4", 3), "<f4", ("<f4", 3), "<f4", "<i4", "<i4", "<f4", "<i4", "<u4", ("<u2", 6)],
        # rng: the per-ray RNG stream word in Ray::data[64..67]; known: the instances (+1) the ray has crossed without a hit on its current
        # segment, Ray::data[68..79] (the schedulers' known-miss shortcut) -- both unused by the reference
        "offsets": [0, 12, 16, 28, 32, 44, 48, 52, 56, 60, 64, 68],
        "itemsize": 80,
    }
)

MATERIAL_DTYPE = np.dtype(
    {
        "names": ["type", "ka", "ks", "kd", "alpha", "eta", "k", "roughness", "hsc", "backScattering", "hsFallOff"],
        "formats": ["<i4", ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", ("<f4", 3), ("<f4", 3), "<f4", ("<f4", 3), "<f4", "<f4"],
        "offsets": [0, 4, 16, 28, 40, 44, 56, 68, 72, 84, 88],
        "itemsize": 92,
    }
)

LIGHT_DTYPE = np.dtype(
    {
        "names": ["type", "position", "color", "normal", "width", "height"],
        "formats": ["<i4", ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", "<f4"],
        "offsets": [0, 4, 16, 28, 40, 44],
        "itemsize": 64,
    }
)

HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<i4"), ("u", "<f4"), ("v", "<f4")])


def default_material(kd=(0.5, 0.5, 0.5), mtype=LAMBERT, ks=(0.5, 0.5, 0.5), alpha=1.0):
    """Material() defaults, Material.h:62-77."""
    m = np.zeros(1, MATERIAL_DTYPE)
    m["type"] = mtype
    m["kd"] = kd
    m["ks"] = ks
    m["alpha"] = alpha
    m["eta"] = (0.19, 1.45, 1.50)
    m["k"] = (3.06, 2.40, 1.88)
    m["roughness"] = 0.05
    return m


def point_light(pos, color=(1.0, 1.0, 1.0)):
    l = np.zeros(1, LIGHT_DTYPE)
    l["type"] = LIGHT_POINT
    l["position"] = pos
    l["color"] = color
    return l


def area_light(pos, color, normal, width, height):
    l = np.zeros(1, LIGHT_DTYPE)
    l["type"] = LIGHT_AREA
    l["position"] = pos
    l["color"] = color
    l["normal"] = normal
    l["width"] = width
    l["height"] = height
    return l


def ambient_light(color=(1.0, 1.0, 1.0)):
    l = np.zeros(1, LIGHT_DTYPE)
    l["type"] = LIGHT_AMBIENT
    l["color"] = color
    return l
