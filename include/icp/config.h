// Drop-in for cornellev/icp include/icp/config.h:9-33 — same class, same semantics:
// string -> variant<int, double, string>; get<T> on another alternative throws
// std::bad_variant_access (std::get<T>, config.h:30).
#pragma once

#include <string>
#include <unordered_map>
#include <variant>

namespace icp {
    /** Configuration for ICP instances. */
    class Config {
        using Param = std::variant<int, double, std::string>;
        std::unordered_map<std::string, Param> params;

    public:
        Config() = default;

        template<typename T>
        void set(const std::string& key, T value) {
            params[key] = value;
        }

        template<typename T>
        T get(const std::string& key, T otherwise) const {
            auto it = params.find(key);
            if (it == params.end()) return otherwise;
            return std::get<T>(it->second);
        }

        /** Extension (not in the reference): true when `key` is present. */
        bool has(const std::string& key) const { return params.find(key) != params.end(); }
    };
}
