from .feeding import (DatasetMapper3D, ResizeShortestEdge, build_detection_test_loader, load_omni3d_json,  # noqa: F401
                      merge_oracle2d_to_detection_dicts, read_image)
