"""placeholder package (opencv_from_cameras_projection is fused into csrc/pd_ggs.hip)"""
