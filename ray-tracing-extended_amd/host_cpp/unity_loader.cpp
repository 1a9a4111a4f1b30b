// unity_loader.cpp — reads the reference's .unity scenes (Unity YAML) into the C++ host components.
//
// What is read (SURVEY.md §8b "Scene surface"): components by script GUID (Assets/Scripts/RayTracingManager.cs.meta:2,
// "Assets/Scripts/Render Types/RayTracedSphere.cs.meta":2, "Assets/Scripts/Render Types/RayTracedMesh.cs.meta":2); manager
// settings (e.g. Assets/Scenes/Chess.unity:30174-30185); sphere / mesh materials and the serialised MeshSplitter chunks
// ("Assets/Scenes/Reflective Balls.unity":356-446); Transform hierarchy, PrefabInstance overrides + m_TransformParent
// (Assets/Scenes/Knight.unity:204-263); Camera "field of view"; directional Light rotation.  FBX prefab roots keep the
// importer's scale 100 / rotation -90 deg X unless overridden.  Same logic as unity_scene.py; buffers are byte-identical.
#include "rt_host.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace rthost {
namespace {

// ---- the YAML subset Unity writes ------------------------------------------------------------------------------------
struct Node {
    enum Kind { Scalar, Map, List } kind = Scalar;
    std::string scalar;
    std::vector<std::pair<std::string, std::shared_ptr<Node>>> map;
    std::vector<std::shared_ptr<Node>> list;
    const Node* get(const std::string& k) const
    {
        for (auto& kv : map) if (kv.first == k) return kv.second.get();
        return nullptr;
    }
    const Node& at(const std::string& k) const
    {
        const Node* n = get(k);
        if (!n) throw std::runtime_error("missing key '" + k + "'");
        return *n;
    }
    double num() const { return scalar.empty() ? 0.0 : std::strtod(scalar.c_str(), nullptr); }
    long long integer() const { return scalar.empty() ? 0 : std::strtoll(scalar.c_str(), nullptr, 10); }
};
using NodeP = std::shared_ptr<Node>;

std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

NodeP parse_flow(const std::string& s, size_t& i);
NodeP parse_value(const std::string& raw)
{
    std::string v = trim(raw);
    if (!v.empty() && v[0] == '{') { size_t i = 0; return parse_flow(v, i); }
    auto n = std::make_shared<Node>();
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) v = v.substr(1, v.size() - 2);
    n->scalar = v;
    return n;
}
NodeP parse_flow(const std::string& s, size_t& i)           // { key: value, key: {..} }
{
    auto n = std::make_shared<Node>(); n->kind = Node::Map;
    ++i;                                                    // '{'
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ' ' || s[i] == ',')) ++i;
        if (i < s.size() && s[i] == '}') { ++i; break; }
        size_t c = s.find(':', i);
        if (c == std::string::npos) break;
        std::string key = trim(s.substr(i, c - i));
        i = c + 1;
        while (i < s.size() && s[i] == ' ') ++i;
        if (i < s.size() && s[i] == '{') n->map.emplace_back(key, parse_flow(s, i));
        else {
            size_t e = i;
            while (e < s.size() && s[e] != ',' && s[e] != '}') ++e;
            n->map.emplace_back(key, parse_value(s.substr(i, e - i)));
            i = e;
        }
    }
    return n;
}

struct Line { int indent; std::string text; };

// Block parser over lines[pos..): returns the node made of all lines with indentation >= `indent`.
NodeP parse_block(const std::vector<Line>& L, size_t& pos, int indent);

NodeP parse_list(const std::vector<Line>& L, size_t& pos, int indent)
{
    auto n = std::make_shared<Node>(); n->kind = Node::List;
    while (pos < L.size() && L[pos].indent == indent && L[pos].text.rfind("- ", 0) == 0) {
        std::string rest = L[pos].text.substr(2);
        std::string t = trim(rest);
        if (!t.empty() && t[0] == '{') { n->list.push_back(parse_value(t)); ++pos; continue; }
        size_t c = t.find(':');
        if (c == std::string::npos) { n->list.push_back(parse_value(t)); ++pos; continue; }
        // "- key: value" opens a map item whose further keys sit at indent + 2
        std::vector<Line> item;
        item.push_back({ indent + 2, t });
        size_t q = pos + 1;
        while (q < L.size() && (L[q].indent > indent + 1 || (L[q].indent == indent + 2))) { if (L[q].indent < indent + 2) break; item.push_back(L[q]); ++q; }
        size_t ip = 0;
        n->list.push_back(parse_block(item, ip, indent + 2));
        pos = q;
    }
    return n;
}

NodeP parse_block(const std::vector<Line>& L, size_t& pos, int indent)
{
    if (pos < L.size() && L[pos].text.rfind("- ", 0) == 0) return parse_list(L, pos, L[pos].indent);
    auto n = std::make_shared<Node>(); n->kind = Node::Map;
    while (pos < L.size() && L[pos].indent >= indent) {
        if (L[pos].indent > indent) { ++pos; continue; }                          // stray continuation
        const std::string& t = L[pos].text;
        size_t c = t.find(':');
        if (c == std::string::npos) { ++pos; continue; }
        std::string key = t.substr(0, c), val = trim(t.substr(c + 1));
        ++pos;
        if (!val.empty()) { n->map.emplace_back(key, parse_value(val)); continue; }
        if (pos < L.size() && L[pos].text.rfind("- ", 0) == 0 && L[pos].indent >= indent) n->map.emplace_back(key, parse_list(L, pos, L[pos].indent));
        else if (pos < L.size() && L[pos].indent > indent) n->map.emplace_back(key, parse_block(L, pos, L[pos].indent));
        else n->map.emplace_back(key, std::make_shared<Node>());
    }
    return n;
}

struct Doc { int cls; long long id; bool stripped; NodeP body; };

std::vector<Doc> parse_documents(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<Doc> docs;
    std::vector<Line> lines;
    Doc cur{ 0, 0, false, nullptr };
    bool have = false;
    auto flush = [&]() {
        if (!have) return;
        size_t pos = 0;
        NodeP root = parse_block(lines, pos, 0);
        cur.body = (root->kind == Node::Map && !root->map.empty()) ? root->map[0].second : std::make_shared<Node>();
        docs.push_back(cur);
        lines.clear();
    };
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.rfind("--- !u!", 0) == 0) {
            flush();
            cur = Doc{ 0, 0, false, nullptr };
            const char* p = line.c_str() + 7;
            char* e;
            cur.cls = (int)std::strtol(p, &e, 10);
            const char* amp = std::strchr(e, '&');
            cur.id = amp ? std::strtoll(amp + 1, &e, 10) : 0;
            cur.stripped = line.find(" stripped") != std::string::npos;
            have = true;
            continue;
        }
        if (!have || line.empty() || line[0] == '%') continue;
        size_t ind = line.find_first_not_of(' ');
        if (ind == std::string::npos) continue;
        lines.push_back({ (int)ind, line.substr(ind) });
    }
    flush();
    return docs;
}

// ---- scene assembly -----------------------------------------------------------------------------------------------
const char* GUID_MANAGER = "68c390cdf7a860745bbbdeccd7d206a9";
const char* GUID_SPHERE = "52a9ac6d93ef8ff438ff410be33e635a";
const char* GUID_MESH = "da1318d85859d584682b30dbc26ca9f6";

struct Pose { Vector3 pos; Quaternion rot; Vector3 scale{1, 1, 1}; long long parent = 0; };

Vector3 xyz(const Node& n) { return { (float)n.at("x").num(), (float)n.at("y").num(), (float)n.at("z").num() }; }
void colour(const Node& n, float out[4]) { out[0] = (float)n.at("r").num(); out[1] = (float)n.at("g").num(); out[2] = (float)n.at("b").num(); out[3] = (float)n.at("a").num(); }
void colour(const Node& n, double out[4]) { out[0] = n.at("r").num(); out[1] = n.at("g").num(); out[2] = n.at("b").num(); out[3] = n.at("a").num(); }

RayTracingMaterial material(const Node& n)
{
    RayTracingMaterial m{};
    colour(n.at("colour"), m.colour); colour(n.at("emissionColour"), m.emissionColour); colour(n.at("specularColour"), m.specularColour);
    m.emissionStrength = (float)n.at("emissionStrength").num(); m.smoothness = (float)n.at("smoothness").num();
    m.specularProbability = (float)n.at("specularProbability").num(); m.flag = (int32_t)n.at("flag").integer();
    return m;
}

std::vector<MeshChunk> chunks(const Node* raw)
{
    std::vector<MeshChunk> out;
    if (!raw) return out;
    for (auto& cn : raw->list) {
        MeshChunk c;
        const Node* tl = cn->get("triangles");
        if (tl) for (auto& tn : tl->list) {
            rt_triangle t{};
            const char* keys[6] = { "posA", "posB", "posC", "normalA", "normalB", "normalC" };
            for (int k = 0; k < 6; ++k) { Vector3 v = xyz(tn->at(keys[k])); float* d = t.posA + 3 * k; d[0] = v.x; d[1] = v.y; d[2] = v.z; }
            c.triangles.push_back(t);
        }
        c.bounds.center = xyz(cn->at("bounds").at("m_Center"));
        Vector3 e = xyz(cn->at("bounds").at("m_Extent"));
        c.bounds.size = { e.x * 2.0f, e.y * 2.0f, e.z * 2.0f };
        c.subMeshIndex = (int)cn->at("subMeshIndex").integer();
        out.push_back(std::move(c));
    }
    return out;
}

} // namespace

RayTracingManager LoadUnityScene(const std::string& path, int width, int height)
{
    std::vector<Doc> docs = parse_documents(path);
    std::map<long long, const Doc*> byId;
    for (const Doc& d : docs) byId[d.id] = &d;

    std::map<long long, Pose> poses, prefabPose;
    std::map<long long, long long> goTransform;
    std::map<long long, bool> goActive, prefabActive;
    for (const Doc& d : docs) {
        const Node& b = *d.body;
        if (d.cls == 4 && !d.stripped) {
            Pose p;
            p.pos = xyz(b.at("m_LocalPosition")); p.scale = xyz(b.at("m_LocalScale"));
            const Node& q = b.at("m_LocalRotation");
            p.rot = { (float)q.at("x").num(), (float)q.at("y").num(), (float)q.at("z").num(), (float)q.at("w").num() };
            p.parent = b.at("m_Father").at("fileID").integer();
            poses[d.id] = p;
            goTransform[b.at("m_GameObject").at("fileID").integer()] = d.id;
        } else if (d.cls == 1 && !d.stripped) {
            const Node* a = b.get("m_IsActive");
            goActive[d.id] = a ? a->integer() != 0 : true;
        } else if (d.cls == 1001) {
            Pose p; p.scale = { 100, 100, 100 }; p.rot = { -0.7071068f, 0, 0, 0.7071067f };
            bool active = true;
            const Node& mod = b.at("m_Modification");
            if (const Node* mods = mod.get("m_Modifications")) for (auto& m : mods->list) {
                const std::string pp = m->at("propertyPath").scalar;
                const Node* v = m->get("value");
                auto setc = [&](float* x, float* y, float* z, float* w, char c) { float f = (float)v->num(); (c == 'x' ? *x : c == 'y' ? *y : c == 'z' ? *z : *w) = f; };
                float dummy = 0;
                if (pp.rfind("m_LocalPosition.", 0) == 0 && pp.size() == 17) setc(&p.pos.x, &p.pos.y, &p.pos.z, &dummy, pp[16]);
                else if (pp.rfind("m_LocalRotation.", 0) == 0 && pp.size() == 17) setc(&p.rot.x, &p.rot.y, &p.rot.z, &p.rot.w, pp[16]);
                else if (pp.rfind("m_LocalScale.", 0) == 0 && pp.size() == 14) setc(&p.scale.x, &p.scale.y, &p.scale.z, &dummy, pp[13]);
                else if (pp == "m_IsActive") active = v->integer() != 0;
            }
            p.parent = mod.at("m_TransformParent").at("fileID").integer();
            prefabPose[d.id] = p; prefabActive[d.id] = active;
        }
    }

    auto world = [&](const Pose* pose) -> Transform {
        std::vector<const Pose*> chain;
        const Pose* p = pose;
        while (p) {
            chain.push_back(p);
            long long par = p->parent;
            if (par == 0) break;
            auto it = poses.find(par);
            if (it != poses.end()) { p = &it->second; continue; }
            auto dit = byId.find(par);
            if (dit != byId.end() && dit->second->cls == 4 && dit->second->stripped) {
                auto pit = prefabPose.find(dit->second->body->at("m_PrefabInstance").at("fileID").integer());
                p = pit != prefabPose.end() ? &pit->second : nullptr;
            } else p = nullptr;
        }
        Vector3 wpos; Quaternion wrot; Vector3 wscale{ 1, 1, 1 };
        for (auto it = chain.rbegin(); it != chain.rend(); ++it) {
            const Pose& q = **it;
            Vector3 r = wrot * Vector3{ wscale.x * q.pos.x, wscale.y * q.pos.y, wscale.z * q.pos.z };
            wpos = { wpos.x + r.x, wpos.y + r.y, wpos.z + r.z };
            wrot = wrot * q.rot;
            wscale = { wscale.x * q.scale.x, wscale.y * q.scale.y, wscale.z * q.scale.z };
        }
        Transform t; t.position = wpos; t.rotation = wrot; t.lossyScale = wscale;
        t.localScale = chain.empty() ? Vector3{ 1, 1, 1 } : chain.front()->scale;
        return t;
    };
    auto transformOfGo = [&](long long go, bool& active) -> Transform {
        const Doc* d = byId.at(go);
        if (d->stripped) {
            long long pi = d->body->at("m_PrefabInstance").at("fileID").integer();
            active = prefabActive.count(pi) ? prefabActive[pi] : true;
            return world(&prefabPose.at(pi));
        }
        active = goActive.count(go) ? goActive[go] : true;
        return world(&poses.at(goTransform.at(go)));
    };

    RayTracingManager mgr;
    mgr.width = width; mgr.height = height;
    const Node* managerDoc = nullptr;
    bool haveCamera = false;
    for (const Doc& d : docs) {
        const Node& b = *d.body;
        if (d.cls == 114 && !d.stripped) {
            const Node* script = b.get("m_Script");
            const Node* g = script ? script->get("guid") : nullptr;
            if (!g) continue;
            long long go = b.at("m_GameObject").at("fileID").integer();
            bool active = true;
            if (g->scalar == GUID_MANAGER) managerDoc = &b;
            else if (g->scalar == GUID_SPHERE) {
                Transform t = transformOfGo(go, active);
                if (active) mgr.spheres.push_back({ t, material(b.at("material")) });
            } else if (g->scalar == GUID_MESH) {
                Transform t = transformOfGo(go, active);
                if (!active) continue;
                RayTracedMesh m; m.transform = t;
                for (auto& mn : b.at("materials").list) m.materials.push_back(material(*mn));
                m.localChunks = chunks(b.get("localChunks"));
                const Node* tc = b.get("triangleCount");
                m.triangleCount = tc ? (int)tc->integer() : 0;
                if (m.triangleCount == 0) for (auto& c : m.localChunks) m.triangleCount += (int)c.triangles.size();
                mgr.meshes.push_back(std::move(m));
            }
        } else if (d.cls == 20 && !d.stripped) {
            bool active;
            mgr.camera.transform = transformOfGo(b.at("m_GameObject").at("fileID").integer(), active);
            mgr.camera.fieldOfView = b.at("field of view").num();
            mgr.camera.aspect = (double)width / (double)height;
            haveCamera = true;
        } else if (d.cls == 108 && !d.stripped && b.get("m_Type") && b.at("m_Type").integer() == 1) {
            bool active;
            mgr.light.rotation = transformOfGo(b.at("m_GameObject").at("fileID").integer(), active).rotation;
        }
    }
    if (!managerDoc || !haveCamera) throw std::runtime_error(path + ": no RayTracingManager / Camera found");
    const Node& M = *managerDoc;
    mgr.maxBounceCount = (int)M.at("maxBounceCount").integer(); mgr.numRaysPerPixel = (int)M.at("numRaysPerPixel").integer();
    mgr.defocusStrength = M.at("defocusStrength").num(); mgr.divergeStrength = M.at("divergeStrength").num();
    mgr.focusDistance = M.at("focusDistance").num();
    const Node& e = M.at("environmentSettings");
    mgr.environmentSettings.enabled = e.at("enabled").integer() != 0;
    colour(e.at("groundColour"), mgr.environmentSettings.groundColour);
    colour(e.at("skyColourHorizon"), mgr.environmentSettings.skyColourHorizon);
    colour(e.at("skyColourZenith"), mgr.environmentSettings.skyColourZenith);
    mgr.environmentSettings.sunFocus = e.at("sunFocus").num(); mgr.environmentSettings.sunIntensity = e.at("sunIntensity").num();
    if (const Node* n = M.get("numMeshChunks")) mgr.serialisedNumMeshChunks = (int)n->integer();
    if (const Node* n = M.get("numTriangles")) mgr.serialisedNumTriangles = (int)n->integer();
    return mgr;
}

} // namespace rthost
