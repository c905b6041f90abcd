// ORACLE / TEST INFRASTRUCTURE: MshIO (un-vendored) reduced to the fields IglUtils.cpp fills and reads.  load_msh
// reads ASCII MSH 4.1 and 2.2 from the published Gmsh grammar and throws on anything else (version 4.0, binary),
// which sends the reference into its own msh-4.0 reader (IglUtils.cpp:462-466) as MshIO does; save_msh writes the
// same 4.1 ASCII sections.
#pragma once
#include <cstddef>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
namespace mshio {
struct MeshFormat {
    std::string version;
    int file_type = 0;
    int data_size = 8;
};
struct NodeBlock {
    int entity_dim = 0, entity_tag = 0, parametric = 0;
    size_t num_nodes_in_block = 0;
    std::vector<size_t> tags;
    std::vector<double> data;
};
struct Nodes {
    size_t num_entity_blocks = 0, num_nodes = 0, min_node_tag = 0, max_node_tag = 0;
    std::vector<NodeBlock> entity_blocks;
};
struct ElementBlock {
    int entity_dim = 0, entity_tag = 0, element_type = 0;
    size_t num_elements_in_block = 0;
    std::vector<size_t> data;
};
struct Elements {
    size_t num_entity_blocks = 0, num_elements = 0, min_element_tag = 0, max_element_tag = 0;
    std::vector<ElementBlock> entity_blocks;
};
struct MshSpec {
    MeshFormat mesh_format;
    Nodes nodes;
    Elements elements;
};
inline int nodes_of_type(int t)
{
    switch (t) {
    case 1: return 2; // line
    case 2: return 3; // triangle
    case 3: return 4; // quad
    case 4: return 4; // tetrahedron
    case 5: return 8; // hexahedron
    case 15: return 1; // point
    default: throw std::runtime_error("refshim mshio: unsupported element type");
    }
}
inline MshSpec load_msh(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("refshim mshio: cannot open " + path);
    MshSpec spec;
    std::string tok;
    auto skip_to_end = [&](const std::string& name) {
        std::string t;
        while (in >> t)
            if (t == "$End" + name) return;
    };
    while (in >> tok) {
        if (tok == "$MeshFormat") {
            in >> spec.mesh_format.version >> spec.mesh_format.file_type >> spec.mesh_format.data_size;
            if (spec.mesh_format.file_type != 0) throw std::runtime_error("refshim mshio: binary files are not read");
            if (spec.mesh_format.version != "4.1" && spec.mesh_format.version != "2.2") throw std::runtime_error("refshim mshio: version " + spec.mesh_format.version);
            skip_to_end("MeshFormat");
        }
        else if (tok == "$Nodes" && spec.mesh_format.version == "4.1") {
            Nodes& n = spec.nodes;
            in >> n.num_entity_blocks >> n.num_nodes >> n.min_node_tag >> n.max_node_tag;
            n.entity_blocks.resize(n.num_entity_blocks);
            for (auto& b : n.entity_blocks) {
                in >> b.entity_dim >> b.entity_tag >> b.parametric >> b.num_nodes_in_block;
                b.tags.resize(b.num_nodes_in_block);
                for (auto& t : b.tags) in >> t;
                const size_t per = 3 + (b.parametric ? (size_t)b.entity_dim : 0);
                b.data.resize(per * b.num_nodes_in_block);
                for (auto& x : b.data) in >> x;
            }
            skip_to_end("Nodes");
        }
        else if (tok == "$Elements" && spec.mesh_format.version == "4.1") {
            Elements& e = spec.elements;
            in >> e.num_entity_blocks >> e.num_elements >> e.min_element_tag >> e.max_element_tag;
            e.entity_blocks.resize(e.num_entity_blocks);
            for (auto& b : e.entity_blocks) {
                in >> b.entity_dim >> b.entity_tag >> b.element_type >> b.num_elements_in_block;
                b.data.resize((size_t)(nodes_of_type(b.element_type) + 1) * b.num_elements_in_block);
                for (auto& x : b.data) in >> x;
            }
            skip_to_end("Elements");
        }
        else if (tok == "$Nodes") { // 2.2
            Nodes& n = spec.nodes;
            in >> n.num_nodes;
            n.num_entity_blocks = 1;
            n.entity_blocks.resize(1);
            NodeBlock& b = n.entity_blocks[0];
            b.entity_dim = 3;
            b.num_nodes_in_block = n.num_nodes;
            b.tags.resize(n.num_nodes);
            b.data.resize(3 * n.num_nodes);
            for (size_t i = 0; i < n.num_nodes; ++i) in >> b.tags[i] >> b.data[3 * i] >> b.data[3 * i + 1] >> b.data[3 * i + 2];
            n.min_node_tag = 1;
            n.max_node_tag = n.num_nodes;
            skip_to_end("Nodes");
        }
        else if (tok == "$Elements") { // 2.2: one block per run of equal (type, entity)
            Elements& e = spec.elements;
            in >> e.num_elements;
            for (size_t i = 0; i < e.num_elements; ++i) {
                size_t id;
                int type, ntags;
                in >> id >> type >> ntags;
                int ent = 0;
                for (int k = 0; k < ntags; ++k) {
                    int t;
                    in >> t;
                    if (k == 1) ent = t;
                }
                if (e.entity_blocks.empty() || e.entity_blocks.back().element_type != type || e.entity_blocks.back().entity_tag != ent) {
                    e.entity_blocks.emplace_back();
                    e.entity_blocks.back().element_type = type;
                    e.entity_blocks.back().entity_tag = ent;
                    e.entity_blocks.back().entity_dim = (type == 4 || type == 5) ? 3 : ((type == 2 || type == 3) ? 2 : (type == 1 ? 1 : 0));
                }
                ElementBlock& b = e.entity_blocks.back();
                b.data.push_back(id);
                for (int k = 0; k < nodes_of_type(type); ++k) {
                    size_t v;
                    in >> v;
                    b.data.push_back(v);
                }
                ++b.num_elements_in_block;
            }
            e.num_entity_blocks = e.entity_blocks.size();
            skip_to_end("Elements");
        }
        else if (tok.size() > 1 && tok[0] == '$' && tok.compare(0, 4, "$End") != 0) {
            skip_to_end(tok.substr(1)); // sections the reference does not use ($Entities, $PhysicalNames, $Surface ...)
        }
    }
    if (spec.mesh_format.version.empty()) throw std::runtime_error("refshim mshio: no $MeshFormat");
    return spec;
}
inline void save_msh(std::ostream& out, const MshSpec& s)
{
    out << "$MeshFormat\n" << s.mesh_format.version << " " << s.mesh_format.file_type << " " << s.mesh_format.data_size << "\n$EndMeshFormat\n";
    out << std::setprecision(17);
    out << "$Nodes\n" << s.nodes.num_entity_blocks << " " << s.nodes.num_nodes << " " << s.nodes.min_node_tag << " " << s.nodes.max_node_tag << "\n";
    for (const auto& b : s.nodes.entity_blocks) {
        out << b.entity_dim << " " << b.entity_tag << " " << b.parametric << " " << b.num_nodes_in_block << "\n";
        for (size_t t : b.tags) out << t << "\n";
        for (size_t i = 0; i < b.num_nodes_in_block; ++i) out << b.data[3 * i] << " " << b.data[3 * i + 1] << " " << b.data[3 * i + 2] << "\n";
    }
    out << "$EndNodes\n$Elements\n" << s.elements.num_entity_blocks << " " << s.elements.num_elements << " " << s.elements.min_element_tag << " " << s.elements.max_element_tag << "\n";
    for (const auto& b : s.elements.entity_blocks) {
        out << b.entity_dim << " " << b.entity_tag << " " << b.element_type << " " << b.num_elements_in_block << "\n";
        const size_t per = (size_t)nodes_of_type(b.element_type) + 1;
        for (size_t i = 0; i < b.num_elements_in_block; ++i) {
            for (size_t k = 0; k < per; ++k) out << (k ? " " : "") << b.data[per * i + k];
            out << "\n";
        }
    }
    out << "$EndElements\n";
}
} // namespace mshio
