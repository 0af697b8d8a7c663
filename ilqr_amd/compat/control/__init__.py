"""Import-path compatibility: put `ilqr_amd/compat` on PYTHONPATH and the
reference's `from control.iLQRController import iLQR` /
`from control.models.car import CarModel` resolve to this package."""
