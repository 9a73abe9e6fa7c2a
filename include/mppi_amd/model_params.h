/**
 * model_params.h — plain-old-data parameter blocks of the precompiled model instantiations.
 *
 * These are what mppi_set_dynamics_params() / mppi_set_cost_params() (include/mppi_amd.h) take as `const void* pod`.
 * Field order and defaults follow the reference's parameter structs so a user of the reference can memcpy theirs:
 *   mppi_cartpole_dynamics_params   <- CartpoleDynamicsParams        dynamics/cartpole/cartpole_dynamics.cuh:6-36
 *   mppi_cartpole_cost_params       <- CartpoleQuadraticCostParams   cost_functions/cartpole/cartpole_quadratic_cost.cuh:10-23
 *                                      (CostParams<1> base: control_cost_coeff[1], discount; cost_functions/cost.cuh:17-31)
 *   mppi_di_dynamics_params         <- DoubleIntegratorParams        dynamics/double_integrator/di_dynamics.cuh:9-37
 *   mppi_di_circle_cost_params      <- DoubleIntegratorCircleCostParams  cost_functions/double_integrator/double_integrator_circle_cost.cuh:8-23
 *   mppi_racer_dubins_params        <- RacerDubinsParams             dynamics/racer_dubins/racer_dubins.cuh:67-87
 *   mppi_racer_dubins_elevation_params <- RacerDubinsElevationParams dynamics/racer_dubins/racer_dubins_elevation.cuh:16-60
 *   mppi_racer_dubins_suspension_params <- RacerDubinsElevationSuspensionParams dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cuh:17-66
 *   mppi_quadratic_cost_params_28   <- QuadraticCostTrajectoryParams<RacerDubins, 1>  cost_functions/quadratic_cost/quadratic_cost.cuh:11-63
 * (paths relative to the reference's include/mppi/).
 */
#ifndef MPPI_AMD_MODEL_PARAMS_H_
#define MPPI_AMD_MODEL_PARAMS_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mppi_cartpole_dynamics_params
{
  float cart_mass;   /* 1.0 */
  float pole_mass;   /* 1.0 */
  float pole_length; /* 1.0 */
} mppi_cartpole_dynamics_params;

typedef struct mppi_cartpole_cost_params
{
  float control_cost_coeff[1];     /* 10.0 */
  float discount;                  /* 1.0 */
  float cart_position_coeff;       /* 1000 */
  float cart_velocity_coeff;       /* 100 */
  float pole_angle_coeff;          /* 2000 */
  float pole_angular_velocity_coeff; /* 100 */
  float terminal_cost_coeff;       /* 0 */
  float desired_terminal_state[4]; /* {0, 0, pi, 0} */
} mppi_cartpole_cost_params;

typedef struct mppi_di_dynamics_params
{
  float system_noise; /* 1.0 */
} mppi_di_dynamics_params;

typedef struct mppi_di_circle_cost_params
{
  float control_cost_coeff[2];    /* {0.01, 0.01} */
  float discount;                 /* 1.0 */
  float velocity_cost;            /* 1 */
  float crash_cost;               /* 1000 */
  float velocity_desired;         /* 2 */
  float inner_path_radius2;       /* 1.875^2 */
  float outer_path_radius2;       /* 2.125^2 */
  float angular_momentum_desired; /* 2 * velocity_desired */
} mppi_di_circle_cost_params;

typedef struct mppi_racer_dubins_params
{
  float c_t[3];                    /* {1.3, 2.6, 3.9} */
  float c_b[3];                    /* {2.5, 3.5, 4.5} */
  float c_v[3];                    /* {3.7, 4.7, 5.7} */
  float c_0;                       /* 4.9 */
  float steering_constant;         /* 0.6 */
  float steer_command_angle_scale; /* 5 */
  float steer_angle_scale;         /* -9.1 */
  float max_steer_angle;           /* 0.5 */
  float max_steer_rate;            /* 5 */
  float steer_accel_constant;      /* 12.1 */
  float steer_accel_drag_constant; /* 1.0 */
  float brake_delay_constant;      /* 6.6 */
  float brake_delay_constant_neg;  /* 8.2 */
  float max_brake_rate_neg;        /* 0.9 */
  float max_brake_rate_pos;        /* 0.33 */
  float wheel_base;                /* 0.3 */
  float low_min_throttle;          /* 0.13 */
  float gravity;                   /* -9.81 */
  int gear_sign;                   /* 1 */
} mppi_racer_dubins_params;

/** RacerDubinsElevationParams: the RacerDubins block followed by the acceleration clamp and the coefficients of the
 *  covariance propagation (feedback gains K_*, process-noise coefficients Q_*) */
typedef struct mppi_racer_dubins_elevation_params
{
  mppi_racer_dubins_params base;
  float clamp_ax;         /* 5.5 */
  float K_x;              /* 1 */
  float K_y;              /* 1 */
  float K_yaw;            /* 1 */
  float K_vel_x;          /* 1 */
  float Q_x_acc;          /* 1 */
  float Q_x_v[3];         /* {41.74219, -0.8187027, -2.2131343} */
  float Q_y_f;            /* 0.1 */
  float Q_omega_v;        /* 0.001 */
  float Q_omega_steering; /* 0 */
} mppi_racer_dubins_elevation_params;

/** RacerDubinsElevationSuspensionParams (dynamics/racer_dubins/racer_dubins_elevation_suspension_lstm.cuh:17-66): the
 *  elevation block followed by the spring / damper suspension */
typedef struct mppi_racer_dubins_suspension_params
{
  mppi_racer_dubins_elevation_params elevation;
  float spring_k;     /* 14000 N/m */
  float drag_c;       /* 1000 N s/m */
  float mass;         /* 1447 kg */
  float I_xx;         /* mass / 12 * 2 * 1.5^2 */
  float I_yy;         /* mass / 12 * (1.5^2 + 3^2) */
  float wheel_radius; /* 0.32 m */
  float c_g[3];       /* {2.981 / 2, 0, 0} */
} mppi_racer_dubins_suspension_params;

/** RacerDubinsElevationUncertaintyParams (dynamics/racer_dubins/racer_dubins_elevation_lstm_unc.cuh:5-48) */
typedef struct mppi_racer_dubins_uncertainty_params
{
  mppi_racer_dubins_suspension_params suspension;
  float unc_scale[7];        /* 1 */
  float pos_quad_brake_c[3]; /* {2.0, 0.5, 0.3} */
  float neg_quad_brake_c[3]; /* {5.84, 0.15, 1.7} */
  int use_static_settling;   /* 1 (the reference's bool) */
} mppi_racer_dubins_uncertainty_params;

/** QuadraticCost over the 28 outputs of the RACER models, one goal (SIM_TIME_HORIZON = 1) */
typedef struct mppi_quadratic_cost_params_28
{
  float control_cost_coeff[2]; /* {0, 0} */
  float discount;              /* 1.0 */
  float s_goal[28];            /* 0 */
  float s_coeffs[28];          /* 1 */
  int current_time;            /* 0 */
} mppi_quadratic_cost_params_28;

/** <- ARStandardCostParams  cost_functions/autorally/ar_standard_cost.cuh:14-41 (float3 r_c1, r_c2, trs as 3 floats each).
 *  The AutoRally NeuralNetModel has no dynamics parameter block (NNDynamicsParams is empty); its weights are the
 *  "dynamics_weights" blob and the costmap is the "costmap" blob of mppi_set_model_blob(). */
typedef struct mppi_ar_standard_cost_params
{
  float control_cost_coeff[2]; /* {0, 0} */
  float discount;              /* 1.0 */
  float desired_speed;         /* 6.0 */
  float speed_coeff;           /* 4.25 */
  float track_coeff;           /* 200 */
  float max_slip_ang;          /* 1.25 */
  float slip_coeff;            /* 10 */
  float track_slop;            /* 0 */
  float crash_coeff;           /* 10000 */
  float boundary_threshold;    /* 0.65 */
  int grid_res;                /* 10 */
  float r_c1[3];               /* world -> texture transform, column 1 */
  float r_c2[3];               /* column 2 */
  float trs[3];                /* translation */
} mppi_ar_standard_cost_params;

#ifdef __cplusplus
}
#endif

#endif /* MPPI_AMD_MODEL_PARAMS_H_ */
