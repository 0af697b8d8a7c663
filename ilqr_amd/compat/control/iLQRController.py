from ilqr_amd.iLQRController import iLQR  # noqa: F401
