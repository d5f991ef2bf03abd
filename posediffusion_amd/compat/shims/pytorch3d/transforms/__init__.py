"""placeholder package (the drop-in util modules compute rotations on the HIP engine)"""
