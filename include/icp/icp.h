// Drop-in for cornellev/icp include/icp/icp.h:49-180 on top of the icpx C ABI.
//
// Same public surface — from_method / is_method_registered / registered_methods, begin, iterate,
// calculate_cost, current_transform, get_matches, ICP2 / ICP3 — and the same protected surface
// third-party subclasses use (Match, transform, a, b, matches, setup; book/writing_icp.md:20-39).
// The built-in methods (impl/*.h) keep their state on the GPU and override the protected
// device_*() hooks, so cost / transform / matches are fetched from the device only when asked for.
#pragma once

#include <cmath>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "icp/config.h"
#include "icp/geo.h"
#include "icpx/icpx.h"

namespace icp {

    template<const Dimension Dim>
    class ICPDriver;

    template<const Dimension Dim>
    class ICP {
    private:
        using MethodConstructor = std::function<std::unique_ptr<ICP<Dim>>(const Config&)>;
        // lib/icp/icp.cpp:17-27 defines these in a translation unit; here they are header-only
        static std::unordered_map<std::string, MethodConstructor>& methods();

    protected:
        using Vector = icp::Vector<Dim>;
        using RBTransform = icp::RBTransform<Dim>;
        using PointCloud = icp::PointCloud<Dim>;

        /** A matching between `point` and `pair` at (arbitrary) cost `cost`. icp.h:61-70 */
        struct Match {
            icp::Index point;
            icp::Index pair;
            double cost;
        };

        mutable RBTransform transform = RBTransform::Identity();
        PointCloud a;
        PointCloud b;
        mutable std::vector<Match> matches;

        ICP(): a(Dim, 0), b(Dim, 0) {}

        virtual void setup() = 0;

        // --- hooks for device-backed methods (no counterpart in the reference) ---
        /** Returns true and fills `out` when the cost lives on the device. */
        virtual bool device_cost(double& out) const { (void)out; return false; }
        /** Brings `transform` up to date from the device. */
        virtual void device_sync_transform() const {}
        /** Brings `matches` up to date from the device. */
        virtual void device_sync_matches() const {}
        /** The icpx handle behind this instance, or nullptr for a host-side subclass. */
        virtual icpx_handle device_handle() const { return nullptr; }
        /** ICPDriver::converge ran the whole loop on the device: adopt its inputs and final transform.
         *  A subclass that returns a device_handle() overrides this (DeviceICP does). */
        virtual void device_adopt_converged(const PointCloud& source, const PointCloud& target,
            const double* final_transform) {
            (void)source; (void)target; (void)final_transform;
        }

        friend class ICPDriver<Dim>;

    public:
        static std::optional<std::unique_ptr<ICP<Dim>>> from_method(const std::string& name,
            const Config& config) {
            auto& m = methods();
            auto it = m.find(name);
            if (it == m.end()) return {};
            return it->second(config);
        }

        static bool is_method_registered(const std::string& name) {
            return methods().find(name) != methods().end();
        }

        static std::vector<std::string> registered_methods() {
            std::vector<std::string> keys;
            for (auto& kv : methods()) keys.push_back(kv.first);
            return keys;
        }

        virtual ~ICP() = default;

        /** icp.h:120-133 */
        void begin(const PointCloud& source, const PointCloud& target, const RBTransform& guess) {
            this->transform = guess;
            this->a = source;
            this->b = target;
            matches.assign(static_cast<size_t>(this->a.cols()), Match{0, 0, 0.0});
            setup();
        }

        virtual void iterate() = 0;

        /** icp.h:149-155 */
        [[nodiscard]] double calculate_cost() const {
            double dev = 0.0;
            if (device_cost(dev)) return dev;
            double sum_squares = 0.0;
            for (auto& match : matches) sum_squares += match.cost;
            return std::sqrt(sum_squares / static_cast<double>(a.cols()));
        }

        /** icp.h:158-160 */
        [[nodiscard]] RBTransform current_transform() const {
            device_sync_transform();
            return transform;
        }

        /** icp.h:167-169 */
        [[nodiscard]] const std::vector<Match>& get_matches() const {
            device_sync_matches();
            return matches;
        }
    };

    using ICP2 = ICP<Dimension::TwoD>;
    using ICP3 = ICP<Dimension::ThreeD>;

    /** Thrown when the HIP library reports an error (the reference has no failing calls here). */
    class IcpxError : public std::runtime_error {
    public:
        int code;
        IcpxError(int c, const char* what): std::runtime_error(what), code(c) {}
    };

    inline void icpx_check(int rc) {
        if (rc != ICPX_OK) throw IcpxError(rc, icpx_last_error());
    }
}

#include "icp/impl/device_icp.h"
#include "icp/impl/feature_aware.h"
#include "icp/impl/trimmed.h"
#include "icp/impl/trimmed_3d.h"
#include "icp/impl/vanilla.h"
#include "icp/impl/vanilla_3d.h"

namespace icp {
    // lib/icp/icp.cpp:17-21
    template<>
    inline std::unordered_map<std::string, ICP2::MethodConstructor>& ICP2::methods() {
        static std::unordered_map<std::string, MethodConstructor> m = {
            {"vanilla", [](const Config& c) { return std::make_unique<Vanilla>(c); }},
            {"trimmed", [](const Config& c) { return std::make_unique<Trimmed>(c); }},
            {"feature_aware", [](const Config& c) { return std::make_unique<FeatureAware>(c); }},
        };
        return m;
    }
    // lib/icp/icp.cpp:24-27
    template<>
    inline std::unordered_map<std::string, ICP3::MethodConstructor>& ICP3::methods() {
        static std::unordered_map<std::string, MethodConstructor> m = {
            {"vanilla", [](const Config& c) { return std::make_unique<Vanilla3d>(c); }},
            {"trimmed", [](const Config& c) { return std::make_unique<Trimmed3d>(c); }},
        };
        return m;
    }
}
