from ilqr_amd.models.Quadrotor_Quat import QuadrotorQuatModel  # noqa: F401
