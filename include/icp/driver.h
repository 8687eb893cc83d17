// Drop-in for cornellev/icp include/icp/driver.h:20-236.
//
// Same class, same setters, same ConvergenceState.  converge() on a built-in (device-backed)
// method is ONE icpx_converge call: begin, the iterate loop and should_terminate run on the GPU.
// For any other ICP subclass it runs the reference's host loop unchanged.
#pragma once

#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <utility>

#include "icp/geo.h"
#include "icp/icp.h"

namespace icp {

    template<const Dimension Dim>
    class ICPDriver {
    public:
        /** driver.h:24-33 */
        struct ConvergenceState {
            double cost = 0;
            size_t iteration_count = 0;
            RBTransform<Dim> transform = RBTransform<Dim>::Identity();
        };

        ICPDriver(std::unique_ptr<ICP<Dim>> icp): icp_(std::move(icp)) {}

        /** driver.h:51-72 */
        ConvergenceState converge(const PointCloud<Dim>& source, const PointCloud<Dim>& target,
            RBTransform<Dim> guess) {
            if (icpx_handle h = icp_->device_handle()) {
                icpx_criteria c{};
                if (min_iterations_) { c.has |= ICPX_HAS_MIN_ITER; c.min_iterations = *min_iterations_; }
                if (max_iterations_) { c.has |= ICPX_HAS_MAX_ITER; c.max_iterations = *max_iterations_; }
                if (stop_cost_) { c.has |= ICPX_HAS_STOP_COST; c.stop_cost = *stop_cost_; }
                if (relative_cost_tolerance_) { c.has |= ICPX_HAS_REL_TOL; c.relative_cost_tolerance = *relative_cost_tolerance_; }
                if (absolute_cost_tolerance_) { c.has |= ICPX_HAS_ABS_TOL; c.absolute_cost_tolerance = *absolute_cost_tolerance_; }
                if (angle_tolerance_rad_ && translation_tolerance_) {
                    c.has |= ICPX_HAS_TRANSFORM_TOL;
                    c.angle_tolerance_rad = *angle_tolerance_rad_;
                    c.translation_tolerance = *translation_tolerance_;
                }
                if (time_limit_) { c.has |= ICPX_HAS_TIME_LIMIT; c.time_limit_s = time_limit_->count(); }
                icpx_state st{};
                icpx_check(icpx_converge(h, source.data(), source.cols(), target.data(), target.cols(),
                    guess.matrix().data(), &c, &st));
                icp_->device_adopt_converged(source, target, st.transform);
                ConvergenceState state;
                state.cost = st.cost;
                state.iteration_count = static_cast<size_t>(st.iteration_count);
                for (int i = 0; i < (Dim + 1) * (Dim + 1); i++) state.transform.matrix().data()[i] = st.transform[i];
                return state;
            }
            return converge_on_host(source, target, guess);
        }

        void set_min_iterations(uint64_t min_iterations) {
            assert(!max_iterations_ || min_iterations <= max_iterations_.value());
            min_iterations_ = min_iterations;
        }
        void set_max_iterations(uint64_t max_iterations) {
            assert(!min_iterations_ || max_iterations >= min_iterations_.value());
            max_iterations_ = max_iterations;
        }
        void set_stop_cost(double stop_cost) { stop_cost_ = stop_cost; }
        void set_relative_cost_tolerance(double v) { relative_cost_tolerance_ = v; }
        void set_absolute_cost_tolerance(double v) { absolute_cost_tolerance_ = v; }
        void set_transform_tolerance(double angle_tolerance, double translation_tolerance) {
            angle_tolerance_rad_ = angle_tolerance;
            translation_tolerance_ = translation_tolerance;
        }
        void set_time_limit(std::chrono::duration<double> time_limit) { time_limit_ = time_limit; }

    private:
        // the reference's loop, for ICP subclasses that do not live on the device
        ConvergenceState converge_on_host(const PointCloud<Dim>& source, const PointCloud<Dim>& target,
            RBTransform<Dim> guess) {
            start_time_ = std::chrono::steady_clock::now();
            icp_->begin(source, target, guess);
            ConvergenceState state;
            state.iteration_count = 0;
            state.cost = icp_->calculate_cost();
            state.transform = icp_->current_transform();
            std::optional<ConvergenceState> last_state{};
            while (!should_terminate(state, last_state)) {
                last_state = state;
                icp_->iterate();
                state.iteration_count++;
                state.cost = icp_->calculate_cost();
                state.transform = icp_->current_transform();
            }
            return state;
        }

        static double rotation_angle(const RBTransform<Dim>& cur, const RBTransform<Dim>& last) {
            // angle of R_cur * R_last^T (driver.h:196-210): Rotation2Dd(...).smallestAngle() in 2D, AngleAxisd(...).angle()
            // in 3D, which goes through a quaternion — the same operations, in the same order, as the device code
            // (icp_amd/csrc/icpx_math.h rotation_angle), so a host-side subclass stops at the same iteration
            const double* c = cur.matrix().data();
            const double* l = last.matrix().data();
            constexpr int L = Dim + 1;
            constexpr double kPi = 3.141592653589793238462643383279502884;
            if constexpr (Dim == Dimension::TwoD) {
                double m00 = c[0 * L + 0] * l[0 * L + 0] + c[1 * L + 0] * l[1 * L + 0];
                double m10 = c[0 * L + 1] * l[0 * L + 0] + c[1 * L + 1] * l[1 * L + 0];
                double a = std::fmod(std::atan2(m10, m00), 2.0 * kPi);
                if (a > kPi) a -= 2.0 * kPi;
                else if (a < -kPi) a += 2.0 * kPi;
                return a;
            } else {
                double m[3][3];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) {
                        double s = c[0 * L + i] * l[0 * L + j];
                        for (int k = 1; k < 3; k++) s += c[k * L + i] * l[k * L + j];
                        m[i][j] = s;
                    }
                double q[4];  // x, y, z, w
                double t = m[0][0] + m[1][1] + m[2][2];
                if (t > 0.0) {
                    t = std::sqrt(t + 1.0);
                    q[3] = 0.5 * t;
                    t = 0.5 / t;
                    q[0] = (m[2][1] - m[1][2]) * t;
                    q[1] = (m[0][2] - m[2][0]) * t;
                    q[2] = (m[1][0] - m[0][1]) * t;
                } else {
                    int i = 0;
                    if (m[1][1] > m[0][0]) i = 1;
                    if (m[2][2] > m[i][i]) i = 2;
                    const int j = (i + 1) % 3, k = (j + 1) % 3;
                    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
                    q[i] = 0.5 * t;
                    t = 0.5 / t;
                    q[3] = (m[k][j] - m[j][k]) * t;
                    q[j] = (m[j][i] + m[i][j]) * t;
                    q[k] = (m[k][i] + m[i][k]) * t;
                }
                const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                return nrm != 0.0 ? 2.0 * std::atan2(nrm, std::abs(q[3])) : 0.0;
            }
        }

        bool should_terminate(ConvergenceState current_state, std::optional<ConvergenceState> last_state) {
            // driver.h:154-222
            if (min_iterations_ && current_state.iteration_count < min_iterations_.value()) return false;
            if (max_iterations_ && current_state.iteration_count >= max_iterations_.value()) return true;
            if (stop_cost_ && current_state.cost < stop_cost_.value()) return true;
            if (time_limit_) {
                auto current_time = std::chrono::steady_clock::now();
                if (current_time - start_time_ > time_limit_.value()) return true;
            }
            if (!last_state) return false;
            double delta_cost = current_state.cost - last_state.value().cost;
            if (absolute_cost_tolerance_ && std::abs(delta_cost) < absolute_cost_tolerance_.value()) return true;
            double relative_cost_change = std::abs(delta_cost) / current_state.cost;
            if (relative_cost_tolerance_ && relative_cost_change < relative_cost_tolerance_.value()) return true;
            if (angle_tolerance_rad_ && translation_tolerance_) {
                double angle_diff = rotation_angle(current_state.transform, last_state.value().transform);
                const double* c = current_state.transform.matrix().data();
                const double* l = last_state.value().transform.matrix().data();
                double tn = 0;
                for (int i = 0; i < Dim; i++) {
                    double d = c[Dim * (Dim + 1) + i] - l[Dim * (Dim + 1) + i];
                    tn += d * d;
                }
                if (std::abs(angle_diff) < angle_tolerance_rad_.value() && std::sqrt(tn) < translation_tolerance_.value())
                    return true;
            }
            return false;
        }

        std::unique_ptr<ICP<Dim>> icp_;
        std::optional<uint64_t> min_iterations_;
        std::optional<uint64_t> max_iterations_;
        std::optional<double> stop_cost_;
        std::optional<double> relative_cost_tolerance_;
        std::optional<double> absolute_cost_tolerance_;
        std::optional<double> angle_tolerance_rad_;
        std::optional<double> translation_tolerance_;
        std::optional<std::chrono::duration<double>> time_limit_;
        std::chrono::time_point<std::chrono::steady_clock> start_time_;
    };
}
