"""Overlay rendering of the tile CLI (reference misc/viz_utils.py:187-214 `visualize_instances_dict_orig`,
called at infer/tile.py:251-257).  Pure host code: contours come from the GPU (cerb_inst_contour_*), drawing is PIL
(`cv2.drawContours` is not available in this image; line rasterisation may differ from OpenCV's by a pixel -- cosmetic)."""

import numpy as np

# colours / line widths of the reference's dataset.yml `viz_info` blocks (configuration data): RGB of the first three entries
DEFAULT_VIZ_INFO = {
    "gland": {"line_width": 12, "inst_colour": (255, 255, 0), "type_colour": {0: (0, 0, 0), 1: (255, 255, 0), 2: (177, 52, 235)}},
    "lumen": {"line_width": 12, "inst_colour": (255, 0, 255), "type_colour": {0: (0, 0, 0), 1: (131, 235, 52)}},
    "nuclei": {"line_width": 3, "inst_colour": (0, 255, 0),
               "type_colour": {0: (0, 0, 0), 1: (0, 0, 255), 2: (0, 255, 0), 3: (255, 0, 255), 4: (176, 244, 230), 5: (0, 191, 255), 6: (255, 165, 0)}},
}


def visualize_instances_dict_orig(input_image, inst_dict_, viz_info=None):
    """input_image: uint8 (H, W, 3) RGB; inst_dict_: {'Gland'|'Lumen'|'Nuclei': {id: {'contour': (K,2) int (x, y), 'type'?: int}}}.
    Draws in the reference's fixed order Gland, Lumen, Nuclei; colour = type colour when the instance has a type, else the
    tissue's instance colour.  Returns a new uint8 RGB array."""
    from PIL import Image, ImageDraw

    viz_info = DEFAULT_VIZ_INFO if viz_info is None else viz_info
    im = Image.fromarray(np.ascontiguousarray(np.asarray(input_image).astype(np.uint8)))
    draw = ImageDraw.Draw(im)
    for tissue in ("Gland", "Lumen", "Nuclei"):
        if tissue not in inst_dict_:
            continue
        vi = viz_info[tissue.lower()]
        width = int(vi["line_width"])
        for _, info in inst_dict_[tissue].items():
            contour = info.get("contour")
            if contour is None or len(contour) < 2:
                continue
            colour = vi["type_colour"].get(int(info["type"]), vi["inst_colour"]) if "type" in info else vi["inst_colour"]
            pts = [(int(x), int(y)) for x, y in np.asarray(contour).reshape(-1, 2)]
            draw.line(pts + [pts[0]], fill=tuple(int(c) for c in colour[:3]), width=width, joint="curve")
    return np.array(im)


def up2_nearest(img):
    """cv2.resize(img, (0, 0), fx=2, fy=2, interpolation=INTER_NEAREST) (infer/tile.py:254)."""
    a = np.asarray(img)
    return np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)
