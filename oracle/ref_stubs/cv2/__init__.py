"""Import stub: src/util/util.py:1,26 evaluates cv2.COLORMAP_HOT as a default argument."""
COLORMAP_HOT = 11
